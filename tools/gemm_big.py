import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from gmf_amd import train as T
dev = torch.device("cuda:0")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K, ta, tb) in ((4096, 4096, 4096, False, True), (4096, 4096, 4096, False, False), (4096, 4096, 4096, True, False), (8192, 8192, 512, False, True)):
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    t1 = timed(lambda: T.gemm(a, b, ta=ta, tb=tb)); t2 = timed(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} ta={ta} tb={tb}: {t1*1e3:.2f} ms {fl/t1/1e12:.1f} TF/s | torch {t2*1e3:.2f} ms {fl/t2/1e12:.1f} TF/s", flush=True)
