import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simdjson_java_amd as S
from oracle import oracle as O
which = sys.argv[1]
mal = gzip.open(os.path.join(ROOT, "tests/golden/data/malformed.txt.gz")).read()
tw = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
docs = {"tw": tw, "mal4400": mal[:4400], "mal4500": mal[:4500], "ff": b'{"a":"abc\xffdef"}' * 50,
        "ctrl": b'{"a":"abc\x01def"}' * 50, "unclosed": b'{"a":"abc' * 50, "ok": b'{"a":"abc"}' * 50}
d = docs[which]
ctx = S.Context(0, 4 << 20)
ctx.set_tile_steps(int(sys.argv[2]))
t = time.time()
idx, st = ctx.stage1(d)
w, ws = O.stage1(d)
print(which, sys.argv[2], "->", idx.size, st, "ok" if (st == ws and np.array_equal(idx, w)) else "MISMATCH", "%.3fs" % (time.time() - t), flush=True)
