"""PointDSC.forward (test mode) captured in a HIP graph (torch.cuda.CUDAGraph) against the eager call: latency at B = 1 and
small batches, and a bitwise check that the replay returns what the eager call returns.  GPU box only."""
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


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, N in ((1, 1000), (1, 5000), (1, 10000), (8, 1000), (32, 1000), (32, 5000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    for _ in range(3):
        ref = model(data)
    eager = timed(lambda: model(data))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(data)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(data)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out["final_trans"], ref["final_trans"]) and torch.equal(out["final_labels"], ref["final_labels"])
    graphed = timed(g.replay)
    print(f"B={B:3d} N={N:6d}: eager {eager:7.3f} ms   graph replay {graphed:7.3f} ms   identical={same}", flush=True)
    del g
