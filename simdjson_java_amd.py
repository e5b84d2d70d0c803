"""Import shim: the package directory is named `simdjson-java_amd` (not a valid Python
identifier), so `import simdjson_java_amd` loads it from that directory."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "simdjson-java_amd")
_spec = importlib.util.spec_from_file_location("simdjson_java_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["simdjson_java_amd"] = _mod
_spec.loader.exec_module(_mod)
