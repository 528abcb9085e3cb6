"""Dev (GPU box): small batches three ways - uniform, the same pairs with their own lengths (N - i) as one ragged call, and pair by pair."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
KEYS = ("corr_pos", "src_keypts", "tgt_keypts")
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
for B, N in ((4, 1000), (8, 1000), (4, 5000), (16, 1000), (2, 5000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    uni = {k: b[k].to(dev) for k in KEYS + ("p_tokens", "q_tokens")}; uni["testing"] = True
    rag = {k: [uni[k][i, :N - i] for i in range(B)] for k in KEYS}; rag.update(p_tokens=uni["p_tokens"], q_tokens=uni["q_tokens"], testing=True)
    one = [{**{k: uni[k][i:i + 1] for k in KEYS + ("p_tokens", "q_tokens")}, "testing": True} for i in range(B)]
    print(f"B={B} N={N}: uniform {t(lambda: model(uni)):.3f} ms, ragged (N - i) {t(lambda: model(rag)):.3f} ms, {B} x B=1 calls {t(lambda: [model(o) for o in one]):.3f} ms")
