"""Achievable HBM rates on this box with plain torch kernels (context for the few-tree numbers)."""
import torch, time
n = 256 * 1024 * 1024  # 1 GiB of float32
a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
t = timeit(lambda: b.copy_(a)); print(f"copy 1 GiB -> 1 GiB: {2 * n * 4 / t / 1e12:.2f} TB/s (read+write)")
t = timeit(lambda: a.sum()); print(f"sum over 1 GiB: {n * 4 / t / 1e12:.2f} TB/s (read)")
c = torch.empty(n // 5, device="cuda")
x5 = a[: (n // 5) * 5].view(-1, 5)
t = timeit(lambda: torch.sum(x5, dim=1, out=c)); print(f"row-sum [N,5] -> [N] (the 24 B/sample pattern): {(n // 5) * 24 / t / 1e12:.2f} TB/s")
# round 5: WRITE-ONLY streams (what the eval kernel's 40 GB of output rows and the Jacobians are): a fill of 4 GiB, plain and through a
# 16-byte-per-lane store pattern like the kernels' (torch's fill uses it already)
big = torch.empty(1024 * 1024 * 1024, device="cuda")  # 4 GiB
t = timeit(lambda: big.fill_(1.5)); print(f"fill 4 GiB: {big.numel() * 4 / t / 1e12:.2f} TB/s (write only)")
t = timeit(lambda: big.zero_()); print(f"memset 4 GiB: {big.numel() * 4 / t / 1e12:.2f} TB/s (write only)")
h = big[: big.numel() // 2]
t = timeit(lambda: torch.add(h, 1.0, out=big[big.numel() // 2:])); print(f"out = in + 1 over 2 GiB -> 2 GiB: {big.numel() * 4 / t / 1e12:.2f} TB/s (read+write)")
