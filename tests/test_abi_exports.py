"""CPU-only: libsjmi.so builds for gfx950, loads, and exports every function include/sjmi.h declares.
No compute call is made (there is no GPU here and no CPU fallback to call)."""
import ctypes
import os
import re

from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    import simdjson_java_amd as S
    S.build()
    lib = ctypes.CDLL(S.lib_path())
    header = open(os.path.join(ROOT, "include", "sjmi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(sjmi_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/sjmi.h but not exported: %s" % missing
    assert set(S.binding.EXPORTS) <= set(names)


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no GPU is present (this container has none)."""
    import pytest
    import simdjson_java_amd as S
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(S.SjmiError):
        S.Context(device=0, capacity=1 << 20)
    with pytest.raises(S.SjmiError):
        S.SimdJsonParser(capacity=1 << 20)


def test_product_does_not_touch_the_oracle():
    """Nothing under simdjson-java_amd/ may import, link or call oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "simdjson-java_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")))
                assert "liboracle" not in code and "sj_oracle" not in code and "import oracle" not in code \
                    and "from oracle" not in code, os.path.join(dirpath, f)
