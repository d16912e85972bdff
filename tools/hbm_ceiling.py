"""Measured HBM ceilings on the box (torch used only as a quick way to launch fill/copy kernels)."""
import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
af = a.view(torch.float32)
print("fill  (write only) GB/s:", n / t(lambda: af.fill_(1.0)) / 1e9)
print("zero  (memset)     GB/s:", n / t(lambda: a.zero_()) / 1e9)
print("copy  (r+w)        GB/s:", 2 * n / t(lambda: b.copy_(a)) / 1e9)
print("sum   (read only)  GB/s:", n / t(lambda: af.sum()) / 1e9)
