"""Latency of one PointDSC.forward (test mode) at the reference's own operating points: B = 1.  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import synthetic                    # noqa: E402

dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
for B, N in ((1, 1000), (1, 5000), (1, 10000), (8, 1000), (32, 1000), (16, 10000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    for _ in range(3):
        model(data)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        model(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d} N={N:6d}: {dt * 1e3:8.3f} ms per forward  = {B * N / dt / 1e6:7.3f} M correspondences/s")

# ---- DGR surface (BASELINE config 5): weighted Procrustes and the robust refinement, N = 8000 ------------------------
import numpy as np                               # noqa: E402

for B in (1, 32, 256):
    N = 8000
    scenes = [synthetic.dgr_scene(N, 500 + i) for i in range(min(B, 32))]
    X = torch.cat([scenes[i % len(scenes)][0] for i in range(B)]).to(dev)
    Y = torch.cat([scenes[i % len(scenes)][1] for i in range(B)]).to(dev)
    w = torch.cat([scenes[i % len(scenes)][2] for i in range(B)]).to(dev)
    off = [i * N for i in range(B + 1)]
    for name, fn in (("weighted_procrustes", lambda: gmf_amd.weighted_procrustes_batched(X, Y, w, off, np.finfo(np.float32).eps)),
                     ("GlobalRegistration ", lambda: gmf_amd.global_registration_batched(X, Y, w, off, break_threshold_ratio=1e-4,
                                                                                         quantization_size=0.1))):
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        extra = f"  (mean {float(out[2][:, 0].mean()):.0f} Adam steps)" if len(out) == 3 else ""
        print(f"DGR {name} B={B:3d} N={N}: {dt * 1e3:8.3f} ms per batch = {B * N / dt / 1e6:8.2f} M correspondences/s{extra}")

# ---- DGR bottleneck fusion (row a15): PerceiverIO 256 / head 128 over M voxels, T = 300 image tokens -------------------
for M in (1000, 4000, 20000, 100000):
    pio = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                              latent_dim_head=64, pe=True).to(dev).eval()
    xq = torch.randn(1, M, 256, device=dev)
    img = torch.randn(1, 300, 128, device=dev)
    for flag in (False, True):                   # the whole layer on the fp32 MFMA | on split-fp16 operands
        pio.split_fp16_ff = pio.split_fp16_attn = flag
        for _ in range(2):
            y = pio(img, queries_encoder=xq)
        torch.cuda.synchronize()
        n, dts = 5, []
        for _ in range(5):                       # median of 5 batches: a one-off stall (allocator, clock ramp) is not the kernel
            t0 = time.perf_counter()
            for _ in range(n):
                y = pio(img, queries_encoder=xq)
            torch.cuda.synchronize()
            dts.append((time.perf_counter() - t0) / n)
        dt = sorted(dts)[2]
        fl = M * (1713152 + 512 * 300)
        print(f"DGR bottleneck PerceiverIO M={M:6d} split_fp16={flag!s:5s}: {dt * 1e3:7.3f} ms  {fl / dt / 1e12:6.1f} TFLOP/s (algorithmic)")
