"""One scene of the randomised parity sweep (tests/tools/parity_sweep.py) under every numerics knob: which part of the HIP path's error
against the fp64 evaluation belongs to which form.  GPU box:  python tests/tools/scene_probe.py [scene index ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import _lib, synthetic
from oracle import gmf_oracle as O

want = [int(a) for a in sys.argv[1:]] or [95]
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
rng = np.random.default_rng(2024)
torch.set_num_threads(16)
h = _lib.handle_for(0)
for s in range(max(want) + 1):
    N = int(rng.choice([64, 200, 333, 500, 777, 1000, 1500, 2048, 3000]))
    T = int(rng.choice([40, 196, 300]))
    if s not in want:
        continue
    b = synthetic.synthetic_batch([1000 + s], N=N, T=T)
    ref = O.pointdsc_forward(sd, b, testing=True)["logits"]
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
    compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
    truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    print(f"scene {s}: N = {N}, T = {T}; logits: |max| {float(truth.abs().max()):.2f}, std {float(truth.std()):.2f}; fp32 oracle vs fp64 {float((ref.double() - truth).abs().max()):.2e}")
    for knobs in ({}, {"pv_fp8": 0}, {"pv_fp8": 0, "fused_linear": 0}, {"compat_cache": 0}, {"scattn_variant": 0}):
        try:
            for k, v in knobs.items():
                h.call("gmf_set_tuning", k.encode(), int(v))
            model(data)
            lg = model.last_logits.cpu()
            e64 = (lg.double() - truth).abs()
            print(f"   {str(knobs):44s} vs fp64 {float(e64.max()):.2e} (mean {float(e64.mean()):.2e})   vs fp32 oracle {float((lg - ref).abs().max()):.2e}   argmax row {int(e64.argmax())}")
        except Exception as ex:
            print("   ", knobs, "->", str(ex)[:100])
        finally:
            for k in knobs:
                h.call("gmf_set_tuning", k.encode(), {"pv_fp8": 1, "fused_linear": 1, "compat_cache": 1, "scattn_variant": 18}.get(k, 0))
