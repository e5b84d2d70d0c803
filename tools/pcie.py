#!/usr/bin/env python
"""PCIe copy rates of this box: H2D / D2H / both at once, from hipHostMalloc'ed (torch pinned) and from
hipHostRegister'ed pageable memory (what the parser's buffers are), 100 MB per copy."""
import time
import numpy as np
import torch

n = 100_000_000
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(n, dtype=torch.uint8).pin_memory()
reg = torch.from_numpy(np.zeros(n, dtype=np.uint8))
reg2 = torch.from_numpy(np.zeros(n, dtype=np.uint8))
rt = torch.cuda.cudart()
for t in (reg, reg2):
    assert int(rt.cudaHostRegister(t.data_ptr(), n, 0)) == 0
page = torch.zeros(n, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, a, b in (("hipHostMalloc", pin, pin2), ("hipHostRegister", reg, reg2), ("pageable", page, None)):
    h2d = timed(lambda: dev.copy_(a, non_blocking=True))
    d2h = timed(lambda: a.copy_(dev, non_blocking=True))
    line = "%-16s H2D %.1f GB/s, D2H %.1f GB/s" % (name, n / h2d / 1e9, n / d2h / 1e9)
    if b is not None:
        def both():
            with torch.cuda.stream(s1):
                dev.copy_(a, non_blocking=True)
            with torch.cuda.stream(s2):
                b.copy_(dev2, non_blocking=True)
        t = timed(both)
        line += ", both directions at once %.1f GB/s each (%.2f ms for 2 x 100 MB)" % (n / t / 1e9, t * 1e3)
    print(line)
