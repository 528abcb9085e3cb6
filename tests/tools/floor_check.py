"""One scene against the fp64 evaluation of the network: the HIP path's error beside the reference's own fp32 error (the floor) and the
floor-relative bound the tests assert (1.5 x floor + 2e-5).  GPU box:  python tests/tools/floor_check.py SEED N T [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import synthetic
from oracle import gmf_oracle as O
seed, N, T = (int(a) for a in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
torch.set_num_threads(16)
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
b = synthetic.synthetic_batch([seed + i for i in range(B)], N=N, T=T)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}; data["testing"] = True
from gmf_amd import _lib
h = _lib.handle_for(0)
ref = O.pointdsc_forward(sd, b, testing=True)["logits"]
b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))
floor = float((ref.double() - truth).abs().max())
for pv in (1, 0):
    h.call("gmf_set_tuning", b"pv_fp8", pv)
    model(data)
    lg = model.last_logits.cpu()
    e64, e32 = float((lg.double() - truth).abs().max()), float((lg - ref).abs().max())
    print(f"seed {seed} B={B} N={N} T={T} pv_fp8={pv}: HIP vs fp64 {e64:.2e}, fp32 oracle vs fp64 (floor) {floor:.2e}, HIP vs oracle {e32:.2e}; "
          f"bound 1.5 x floor + 2e-5 = {1.5 * floor + 2e-5:.2e} -> {'inside' if e64 <= 1.5 * floor + 2e-5 else 'OUTSIDE'}")
h.call("gmf_set_tuning", b"pv_fp8", 1)
