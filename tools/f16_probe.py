"""Dev probe (GPU box): golden F16 scenes - pose error vs ground truth and inlier counts of the HIP pose and the reference's."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "f16_pose_tie_scenes.npz"))
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
m = gmf_amd.PointDSC(num_layers=12); m.load_state_dict(sd, strict=False); m = m.cuda().eval()
def inl(T, b, tau=0.10):
    T = torch.as_tensor(T, dtype=torch.float32)
    p = b["src_keypts"][0] @ T[0, :3, :3].T + T[0, :3, 3]
    return int(((p - b["tgt_keypts"][0]).norm(dim=-1) < tau).sum())
for N, seed in g["cases"]:
    tag = f"{N}_{seed}"
    b = synthetic.synthetic_batch([int(seed)], N=int(N), T=196)
    data = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}; data["testing"] = True
    res = m(data)
    Th = res["final_trans"].cpu().numpy(); Tr = g[f"final_trans_{tag}"]; Tg = g[f"gt_trans_{tag}"]
    print(tag, "err_hip %.3e err_ref %.3e |hip-ref| %.3e" % (np.abs(Th - Tg).max(), np.abs(Tr - Tg).max(), np.abs(Th - Tr).max()),
          "inliers hip/ref/gt", inl(Th, b), inl(Tr, b), inl(Tg, b), "labels ref", int(g[f"final_labels_{tag}"].sum()))
