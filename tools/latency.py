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
