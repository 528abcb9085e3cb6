"""Shape fuzz: HIP path vs the CPU oracle over random (B, N, T) - odd sizes on purpose: every grid plan of the encoder (one kernel per
stage, mixed-role small grids, per-tile roles, key / hidden splits, the two-launch form) is picked by B and N, and a plan that is
wrong for some size only shows at that size.  Uniform batches, then the same pairs as ONE ragged call.
GPU box:  python tests/tools/shape_fuzz.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import synthetic
from oracle import gmf_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
torch.set_num_threads(16)
KEYS = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
worst, bad, t0 = 0.0, [], time.time()
for c in range(n_cases):
    B = int(rng.integers(1, 7))
    N = int(rng.choice([int(rng.integers(64, 400)), int(rng.integers(400, 1400)), int(rng.integers(1400, 2700))]))
    T = int(rng.choice([1, 7, 31, 32, 33, 100, 196, 257, 300]))
    if B * N > 9000: B = max(1, 9000 // N)                     # (keeps the oracle at seconds per case)
    b = synthetic.synthetic_batch([3000 + 10 * c + i for i in range(B)], N=N, T=T)
    ref = O.pointdsc_forward(sd, b, testing=True)
    data = {k: b[k].to(dev) for k in KEYS}; data["testing"] = True
    res = model(data)
    e = float((model.last_logits.cpu() - ref["logits"]).abs().max())
    lab = float((res["final_labels"].cpu() != ref["final_labels"]).float().mean()) if "final_labels" in ref else -1.0
    # the same pairs with DIFFERENT lengths as one ragged call: pair i keeps its first N - 3 i correspondences
    lens = [N - 3 * i for i in range(B)]
    if B > 1 and min(lens) >= 64:
        rag = {k: [b[k][i, :lens[i]].to(dev) for i in range(B)] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        rag.update(p_tokens=data["p_tokens"], q_tokens=data["q_tokens"], testing=True)
        lg = model(rag)["logits"]
        er = 0.0
        for i in range(B):
            bi = {k: (b[k][i:i + 1, :lens[i]] if k in ("corr_pos", "src_keypts", "tgt_keypts", "gt_labels") else b[k][i:i + 1]) for k in b if torch.is_tensor(b[k])}
            ri = O.pointdsc_forward(sd, bi, testing=True)
            er = max(er, float((lg[i].cpu() - ri["logits"][0]).abs().max()))
    else:
        er = float("nan")
    worst = max(worst, e, 0.0 if er != er else er)
    flag = "" if max(e, 0.0 if er != er else er) < 2e-4 else "   <-- LOOK"
    if flag: bad.append((B, N, T, e, er))
    print(f"case {c:3d}: B={B} N={N:5d} T={T:3d}  max|dlogit| uniform {e:.2e}  ragged {er:.2e}  labels differing {lab:.4f}{flag}", flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s: worst {worst:.2e}; flagged: {bad}")
