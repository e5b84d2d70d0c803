"""What this box's HBM gives a plain kernel: device-to-device copy, fill and read of 1 GiB (torch / rocclr kernels), TB/s of bytes moved.
The yardstick beside the traffic the stage-1 kernels move (profiles/r6/README.md, dense stage 1)."""
import time, torch
dev = torch.device("cuda", 0)
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
a.zero_(); b.zero_()
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tc = t(lambda: b.copy_(a)); tf = t(lambda: b.zero_()); tr = t(lambda: a.view(torch.int64).sum())
print("copy 1 GiB: %.3f ms = %.2f TB/s moved (read + write); fill: %.3f ms = %.2f TB/s; read (int64 sum): %.3f ms = %.2f TB/s"
      % (tc * 1e3, 2 * n / tc / 1e12, tf * 1e3, n / tf / 1e12, tr * 1e3, n / tr / 1e12))
