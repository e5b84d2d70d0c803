"""N ranks of a node start at the same moment and each makes sure the checker and the document generator are built.  Round 5: after
sj_oracle.c had changed, eight ranks found liboracle.so stale and rebuilt it AT ONCE; some of them loaded half a file ("file too
short") and `bench.py --gpus 8` died.  The builds now compile into a temporary file and rename it into place under a lock; here eight
processes force a rebuild together and every one of them must load a complete library."""
import os
import subprocess
import sys

from tests.conftest import ROOT

WORKER = """
import sys, ctypes
sys.path.insert(0, %r)
from oracle import oracle
from tools import workloads as W
oracle.build(force=True)
lib = W.build_docgen(force=True)
ctypes.CDLL(oracle._LIB_PATH).sjo_stage1
ctypes.CDLL(oracle._AVX_PATH)
ctypes.CDLL(lib).docgen_fill
print("ok")
""" % ROOT


def test_eight_processes_rebuild_the_helper_libraries_at_once():
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT) for _ in range(8)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and o.decode().strip() == "ok", e.decode()[-2000:]
    # nothing half-written is left behind
    for d in ("oracle", "tools"):
        assert not [f for f in os.listdir(os.path.join(ROOT, d)) if ".tmp." in f]
