"""gmf_gemm_f32 on the shapes one training step of PointDSC issues (16 pairs x 1000 correspondences, 300 tokens): time and
TFLOP/s per shape, against torch.matmul (rocBLAS / hipBLASLt fp32) for orientation.  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmf_amd import train as T                   # noqa: E402

dev = torch.device("cuda:0")
R = 16000
shapes = [
    # (name, ta, tb, M, N, K, batch)
    ("linear  X[R,128] W^T[128,128]", False, True, R, 128, 128, 1),
    ("linear  X[R,128] W^T -> 1024 (FF W1)", False, True, R, 1024, 128, 1),
    ("linear  X[R,512] W^T -> 128 (FF W2)", False, True, R, 128, 512, 1),
    ("dX      dY[R,128] W[128,128]", False, False, R, 128, 128, 1),
    ("dX      dY[R,1024] W[1024,128]", False, False, R, 128, 1024, 1),
    ("dW      dY^T[128,R] X[R,128]", True, False, 128, 128, R, 1),
    ("dW      dY^T[1024,R] X[R,128]", True, False, 1024, 128, R, 1),
    ("dW      dY^T[128,R] X[R,512]", True, False, 128, 512, R, 1),
    ("S=QK^T  [1000,128]x[1000,128]^T x16", False, True, 1000, 1000, 128, 16),
    ("O=PV    [1000,1000]x[1000,128] x16", False, False, 1000, 128, 1000, 16),
    ("dP=dO V^T [1000,128]x[1000,128]^T x16", False, True, 1000, 1000, 128, 16),
    ("dV=P^T dO [1000,1000]^T x [1000,128] x16", True, False, 1000, 128, 1000, 16),
    ("cross S [R,64] x ctx[300,64]^T (per pair 1000x300) x16", False, True, 1000, 300, 64, 16),
]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


g = torch.Generator(device="cpu").manual_seed(0)
for name, ta, tb, M, N, K, batch in shapes:
    a_shape = (K, M) if ta else (M, K)
    b_shape = (N, K) if tb else (K, N)
    if batch > 1:
        a = torch.randn((batch,) + a_shape, generator=g).to(dev)
        b = torch.randn((batch,) + b_shape, generator=g).to(dev)
        out = torch.empty((batch, M, N), device=dev)
        fn = lambda: T.gemm(a, b, ta=ta, tb=tb, out=out, m=M, n=N, k=K, lda=a_shape[1], ldb=b_shape[1], ldc=N, batch=batch,
                            sa=a_shape[0] * a_shape[1], sb=b_shape[0] * b_shape[1], sc=M * N)
        ref = lambda: torch.matmul(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)
    else:
        a = torch.randn(a_shape, generator=g).to(dev)
        b = torch.randn(b_shape, generator=g).to(dev)
        fn = lambda: T.gemm(a, b, ta=ta, tb=tb)
        ref = lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b)
    err = float((fn().reshape(-1) - ref().reshape(-1)).abs().max())
    t1, t2 = timed(fn), timed(ref)
    fl = 2.0 * M * N * K * batch
    print(f"{name:58s} {t1 * 1e6:8.1f} us {fl / t1 / 1e12:6.1f} TF/s | torch {t2 * 1e6:8.1f} us {fl / t2 / 1e12:6.1f} TF/s | maxdiff {err:.1e}",
          flush=True)
