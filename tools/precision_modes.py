"""The two numerics modes of the encoder side by side (GPU box): parity (default: fp32-equivalent split-fp16 products) and
throughput (gmf_set_tuning "precision" = 1: the spatial-consistency attention multiplies plain fp16 operands, c as fp16).
Step time, and the deviation of the throughput mode from the parity mode: logits, inlier labels, final poses.
    python tools/precision_modes.py [B] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import _lib, synthetic              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
kind = sys.argv[3] if len(sys.argv) > 3 else "3dmatch"
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
kw = dict(sigma_d=1.2, inlier_threshold=1.2, nms_radius=1.2) if kind == "kitti" else {}
model = gmf_amd.PointDSC(num_layers=12, **kw)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196, **({"scale": 12.0} if kind == "kitti" and "scale" in synthetic.synthetic_batch.__code__.co_varnames else {}))
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
data["testing"] = True
gt_T = b["gt_trans"].to(dev)
gt_labels = b["gt_labels"].to(dev)
h = _lib.handle_for(0)


def run(mode, n=10):
    h.call("gmf_set_tuning", b"precision", mode)
    for _ in range(3):
        res = model(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        res = model(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt, model.last_logits.clone(), res["final_trans"].clone(), res["final_labels"].clone()


LEVEL = int(os.environ.get("LEVEL", "1"))      # 1: one-product attention + fp16 compat; 2: + one-product linear stages
t0, lg0, T0, lab0 = run(0)
t1, lg1, T1, lab1 = run(LEVEL)
h.call("gmf_set_tuning", b"precision", 0)
print(f"B={B} N={N} {kind} (precision {LEVEL}): parity {t0 * 1e3:.2f} ms = {B * N / t0 / 1e6:.2f} M corr/s | throughput {t1 * 1e3:.2f} ms = {B * N / t1 / 1e6:.2f} M corr/s"
      f"  ({t0 / t1:.2f}x)")
dl = (lg1 - lg0).abs()
print(f"logits: max |d| {float(dl.max()):.3e}, mean |d| {float(dl.mean()):.3e}, logit scale (rms) {float(lg0.pow(2).mean().sqrt()):.2f}")
print(f"sign(logit) agreement {float(((lg1 > 0) == (lg0 > 0)).float().mean()) * 100:.3f} %; final labels agreement "
      f"{float((lab1 == lab0).float().mean()) * 100:.3f} %")
print(f"classification accuracy vs ground truth: parity {float(((lg0 > 0).float() == gt_labels).float().mean()) * 100:.3f} %, "
      f"throughput {float(((lg1 > 0).float() == gt_labels).float().mean()) * 100:.3f} %")
print(f"final_trans: max |T1 - T0| {float((T1 - T0).abs().max()):.3e}; vs ground truth: parity {float((T0 - gt_T).abs().max()):.3e}, "
      f"throughput {float((T1 - gt_T).abs().max()):.3e}")
