"""Soak test (GPU box): the same forward many times over - every result must be bitwise equal to the first (a missed wait or
a racing refill in a hand-scheduled kernel shows up as a flipped bit sooner or later).  Both numerics modes, large, mid-size
and small grids, and a training step's gradients.
    python tools/soak_determinism.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import synthetic                    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
bad = 0
for B, N, mode, n in ((32, 5000, "parity", reps // 4), (32, 5000, "throughput", reps // 4), (33, 1000, "parity", reps), (1, 5000, "parity", reps),
                      (1, 1000, "parity", reps), (3, 3001, "parity", reps // 2), (16, 10000, "throughput", reps // 8)):
    model.set_precision(mode)
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    ref = model(data)
    lg0, T0 = model.last_logits.clone(), ref["final_trans"].clone()
    diff = 0
    for _ in range(max(n, 2)):
        out = model(data)
        if not (torch.equal(model.last_logits, lg0) and torch.equal(out["final_trans"], T0)):
            diff += 1
    bad += diff
    print(f"B={B} N={N} {mode}: {max(n, 2)} repeats, {diff} differ", flush=True)
model.set_precision("parity")
# [r5] the guarded path with layers on BOTH sides of the guard (the seeded KITTI-shape network trips its first five layers), and the
# forward from raw images (native small-grid convolutions, per-tile cross-attention role)
km = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=1.2, sigma_d=1.2, k=40, nms_radius=1.2)
km.load_state_dict(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2), strict=False)
km = km.to(dev).eval()
for B, N, n in ((9, 3970, reps // 4), (1, 2000, reps)):
    b = synthetic.synthetic_batch(list(range(80, 80 + B)), N=N, T=196, kind="kitti")
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    ref = km(data)
    lg0, T0 = km.last_logits.clone(), ref["final_trans"].clone()
    diff = sum(0 if (torch.equal(km(data)["final_trans"], T0) and torch.equal(km.last_logits, lg0)) else 1 for _ in range(max(n, 2)))
    bad += diff
    print(f"KITTI stress weights (guard tripped) B={B} N={N}: {max(n, 2)} repeats, {diff} differ", flush=True)
b = synthetic.synthetic_batch([5], N=1000, T=300)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
data.update(p_image=torch.rand(1, 3, 120, 160, device=dev), q_image=torch.rand(1, 3, 120, 160, device=dev), testing=True)
ref = model(data)
lg0 = model.last_logits.clone()
diff = sum(0 if torch.equal((model(data), model.last_logits)[1], lg0) else 1 for _ in range(reps))
bad += diff
print(f"B=1 N=1000 from raw images: {reps} repeats, {diff} differ", flush=True)
# [r5] ragged batches: a small one (key-split items planned on the smallest pair, the role linear kernel, k_scattn_merge with the pair
# table) and a larger one (whole items in work-balanced slots), each in two orders of the same pairs - the result of a pair must
# not depend on its slot
for sizes, n in (([1000, 2311, 640, 1500], reps), ([3970, 2500, 3100, 3970, 1800, 3333, 2900, 3600, 2050, 3970], reps // 4)):
    prs = [synthetic.synthetic_batch([300 + i], N=nn, T=196) for i, nn in enumerate(sizes)]
    def ragged(order):
        d = {k: [prs[i][k][0].to(dev) for i in order] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        d["p_tokens"] = torch.cat([prs[i]["p_tokens"] for i in order]).to(dev)
        d["q_tokens"] = torch.cat([prs[i]["q_tokens"] for i in order]).to(dev)
        d["testing"] = True
        return d
    fwd, rev = list(range(len(sizes))), list(range(len(sizes)))[::-1]
    r0 = model(ragged(fwd))
    lg0, T0 = [x.clone() for x in r0["logits"]], r0["final_trans"].clone()
    diff = 0
    for it in range(max(n, 2)):
        order = rev if it & 1 else fwd
        r = model(ragged(order))
        same = all(torch.equal(r["logits"][j], lg0[i]) for j, i in enumerate(order)) and torch.equal(r["final_trans"], T0[order])
        diff += 0 if same else 1
    bad += diff
    print(f"ragged {len(sizes)} pairs (n = {min(sizes)} .. {max(sizes)}), both orders: {max(n, 2)} repeats, {diff} differ", flush=True)
# training step: the gradients of two identical steps from identical weights
m = gmf_amd.PointDSC(num_layers=3)
m.load_state_dict(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 3, 128), seed=7), strict=False)
m = m.to(dev).train()
b = synthetic.synthetic_batch([1, 2, 3, 4], N=500, T=40)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
gt = b["gt_labels"].to(dev)
state = {k: v.clone() for k, v in m.state_dict().items()}
grads = []
for _ in range(max(reps // 20, 3)):
    m.load_state_dict(state)
    m.zero_grad()
    res = m(data)
    loss = gmf_amd.ClassificationLoss(balanced=False)(res["final_labels"], gt)["loss"] + gmf_amd.SpectralMatchingLoss(balanced=False)(res["M"], gt)
    loss.backward()
    grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).clone())
d = sum(int(not torch.equal(g, grads[0])) for g in grads[1:])
bad += d
print(f"training step: {len(grads)} repeats, {d} differ", flush=True)
print("DETERMINISTIC" if bad == 0 else f"NON-DETERMINISTIC: {bad}")
