"""A/B timing of the attention kernel variants in ONE process, interleaved rounds (guide rule 24),
plus each variant's logit error against the CPU oracle on one pair.  GPU box only:
    python tests/tools/ab_scattn.py [B] [N]
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd                                   # noqa: E402
from gmf_amd import _lib, synthetic              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1,2".split(","))]
if len(sys.argv) > 4:      # optional extra knob, e.g. h2_double_buffer=0
    knob, val = sys.argv[4].split("=")
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
args = [b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
h = _lib.handle_for(0)
if len(sys.argv) > 4:
    h.call("gmf_set_tuning", knob.encode(), int(val))

ref = None
if N <= 5000:
    from oracle import gmf_oracle as O
    ref = O.pointdsc_forward(sd, {k: v[:1] for k, v in b.items()}, testing=False)["logits"]

times = {v: [] for v in variants}
errs = {}
for rnd in range(4):
    for v in variants:
        h.call("gmf_set_tuning", b"scattn_variant", v)
        h.call("gmf_profile_enable", 1)
        lg = model.encode(*args)[0]
        torch.cuda.synchronize()
        ms, n = C.c_double(0), C.c_int(0)
        h.call("gmf_profile_read", C.byref(ms), C.byref(n))
        h.call("gmf_profile_enable", 0)
        if rnd > 0:
            times[v].append(ms.value / n.value)
        if ref is not None:
            errs[v] = float((lg[:1].cpu() - ref).abs().max())
flops = B * (512.0 * N * N + 40960.0 * N)
for v in variants:
    med = statistics.median(times[v])
    print(f"variant {v}: k_scattn median {med:.3f} ms (min {min(times[v]):.3f})  {flops / med / 1e9:.1f} TFLOP/s "
          f"= {flops / med / 1e9 / 157.3:.3f} of fp32-MFMA peak   max|dlogit| vs oracle = {errs.get(v, float('nan')):.2e}")
