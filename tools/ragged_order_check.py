"""Dev (GPU box): a ragged batch with very different pair sizes - every pair equal to its own B = 1 call, and the time of the launch."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic, _lib
if os.environ.get("GMF_LIB"):                    # A/B against another build of the library
    _lib.LIB_PATH = os.environ["GMF_LIB"]
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
rng = np.random.default_rng(1)
KEYS = ("corr_pos", "src_keypts", "tgt_keypts")
for B, lo, hi in ((32, 4000, 5500), (32, 1000, 6000), (13, 2000, 5000)):
    sizes = [int(v) for v in rng.integers(lo, hi, B)]
    pairs = [synthetic.synthetic_batch([200 + i], N=n, T=196) for i, n in enumerate(sizes)]
    rag = {k: [p[k][0].to(dev) for p in pairs] for k in KEYS}
    rag["p_tokens"] = torch.cat([p["p_tokens"] for p in pairs]).to(dev); rag["q_tokens"] = torch.cat([p["q_tokens"] for p in pairs]).to(dev)
    rag["testing"] = True
    r = model(rag)
    bad = 0
    for i in (0, 1, B // 2, B - 1):
        one = {k: pairs[i][k].to(dev) for k in KEYS + ("p_tokens", "q_tokens")}; one["testing"] = True
        model(one)
        bad = max(bad, float((model.last_logits[0] - r["logits"][i]).abs().max()))
    for _ in range(3): model(rag)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): model(rag)
    torch.cuda.synchronize()
    print(f"B={B} N ~ U[{lo}, {hi}) sum {sum(sizes)} sum n^2 {sum(n * n for n in sizes) / 1e6:.0f} M: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per ragged forward; "
          f"largest difference of four pairs from their own B = 1 call: {bad:.2e}")
