"""Odd and degenerate inputs through the whole forward (GPU box): ragged N, tiny N, duplicate points, a zero-variance scene,
both numerics modes - everything must stay finite and the two seeds-free outputs must not depend on padding."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import synthetic                    # noqa: E402

dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
ok = True
for B, N, T in ((1, 41, 7), (3, 63, 196), (2, 65, 300), (1, 4999, 196), (5, 1001, 33), (40, 1000, 196), (9, 5001, 196)):
    for mode in ("parity", "throughput"):
        model.set_precision(mode)
        b = synthetic.synthetic_batch(list(range(B)), N=N, T=T)
        data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
        data["testing"] = True
        out = model(data)
        fin = bool(torch.isfinite(out["final_trans"]).all() and torch.isfinite(model.last_logits).all())
        ok &= fin
        print(f"B={B} N={N} T={T} {mode}: finite={fin}", flush=True)
model.set_precision("parity")
# degenerate scenes: all points identical; target = source (identity); zero tokens
b = synthetic.synthetic_batch([0, 1], N=300, T=40)
for name, edit in (("identical points", lambda d: d.update(src_keypts=d["src_keypts"] * 0 + 0.5, tgt_keypts=d["tgt_keypts"] * 0 + 0.7)),
                   ("target = source", lambda d: d.update(tgt_keypts=d["src_keypts"].clone())),
                   ("zero tokens", lambda d: d.update(p_tokens=d["p_tokens"] * 0, q_tokens=d["q_tokens"] * 0))):
    data = {k: b[k].to(dev).clone() for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    edit(data)
    data["corr_pos"] = torch.cat([data["src_keypts"], data["tgt_keypts"]], -1)
    data["corr_pos"] = data["corr_pos"] - data["corr_pos"].mean(1, keepdim=True)
    data["testing"] = True
    out = model(data)
    fin = bool(torch.isfinite(out["final_trans"]).all() and torch.isfinite(model.last_logits).all())
    ok &= fin
    print(f"{name}: finite={fin}  T[0]=\n{out['final_trans'][0].cpu().numpy().round(3)}", flush=True)
print("ALL FINITE" if ok else "NON-FINITE OUTPUT")
