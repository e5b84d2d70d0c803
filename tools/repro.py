import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simdjson_java_amd as S
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "malformed.txt"
d = gzip.open(os.path.join(ROOT, "tests/golden/data/%s.gz" % name)).read()
ctx = S.Context(0, 96 << 20)
for steps in (8, 4, 2):
    ctx.set_tile_steps(steps)
    for n in (len(d), 8192, 8193, 16384, 20000):
        print("steps", steps, "len", n, flush=True)
        t = time.time()
        try:
            idx, st = ctx.stage1(d, n)
            w, ws = O.stage1(d, n)
            print("   ->", idx.size, st, "ok" if (st == ws and np.array_equal(idx, w)) else "MISMATCH", "%.3fs" % (time.time() - t), flush=True)
        except Exception as e:
            print("   -> EXC", e, "%.3fs" % (time.time() - t), flush=True)
