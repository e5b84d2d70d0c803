#!/usr/bin/env python
"""Known-byte-count device kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (SURVEY.md 5,
MI355X_MICROARCH.md 'HBM'): run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE, separate passes);
tools/archive/summarize_prof_r2.py compares the counters of these dispatches with the bytes they are known to move.
  copy : dst.copy_(src), 1 GiB of uint8 viewed as int32 (reads 1 GiB, writes 1 GiB; far larger than the 256 MiB L3)
  read : src.sum()        (reads 1 GiB, writes ~nothing)
  fill : dst.fill_(7)     (writes 1 GiB)"""
import torch

N = 1 << 30
src = torch.empty(N // 4, dtype=torch.int32, device="cuda").random_(0, 100)
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
for _ in range(3):
    src.sum()
torch.cuda.synchronize()
for _ in range(3):
    dst.fill_(7)
torch.cuda.synchronize()
print("calibration kernels done: copy/read/fill of %d bytes, 3 launches each" % N)
