import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import simdjson_java_amd as S
from oracle import oracle as O
from tests.conftest import load_fixture
doc = bytes(load_fixture("wide_bench.json"))
c = S.Context(device=0, capacity=64 << 20)
idx, st = c.stage1(doc)
want_sb, want_offs, feo, fec = O.unescape_all(doc + b"\0" * 64, idx)
got_sb, fei, gec = c.unescape(len(doc) + 4 * idx.size + 64)
print(len(want_sb), len(got_sb), fei, gec)
# walk the records
p = 0; k = 0
qpos = [i for i in range(idx.size) if doc[idx[i]] == 0x22]
while p < len(want_sb):
    lw = int.from_bytes(want_sb[p:p+4], "big"); lg = int.from_bytes(got_sb[p:p+4], "big")
    if lw != lg or want_sb[p:p+4+lw] != got_sb[p:p+4+lw]:
        i = qpos[k]
        print("string", k, "structural", i, "lane", i % 64, "group", i // 64, "want len", lw, "got", lg)
        print(doc[idx[i]:idx[i] + lw + 20])
        print(want_sb[p+4:p+4+lw]); print(got_sb[p+4:p+4+max(lw, lg)])
        # neighbours in the same wave group
        g0 = (i // 64) * 64
        for t in range(g0, g0 + 64):
            if doc[idx[t]] == 0x22:
                e = doc.index(b'"', idx[t] + 1)
                seg = doc[idx[t]:idx[t+1]]
                if b"\\" in seg: print("  esc lane", t % 64, len(seg), seg[:70])
        break
    p += 4 + lw; k += 1
