"""Which stage of the pose head makes a final pose differ from the oracle's?  GPU box."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import synthetic
from oracle import gmf_oracle as O
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
rng = np.random.default_rng(2024)
torch.set_num_threads(16)
for s in range(40):
    N = int(rng.choice([64, 200, 333, 500, 777, 1000, 1500, 2048, 3000]))
    T = int(rng.choice([40, 196, 300]))
    b = synthetic.synthetic_batch([1000 + s], N=N, T=T)
    ref = O.pointdsc_forward(sd, b, testing=True)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    te = float((res["final_trans"].cpu() - ref["final_trans"]).abs().max())
    if te < 1e-4:
        continue
    # feed the ORACLE's logits and features to the HIP pose head: does the difference come from the encoder or the head?
    feat_n = torch.nn.functional.normalize(ref["corr_features"], p=2, dim=-1).to(dev).contiguous()
    T2, lab2, aux = model.pose_head(feat_n, data["src_keypts"], data["tgt_keypts"], ref["logits"].to(dev).contiguous(), True, return_aux=True)
    te2 = float((T2.cpu() - ref["final_trans"]).abs().max())
    seeds_ref = ref["seeds"][0].numpy()
    seeds_hip = aux["seeds"][0].cpu().numpy()
    # and the HIP logits/features through the ORACLE's pose head
    lg = model.last_logits.cpu(); fn = model.last_features.cpu()
    S = int(N * 0.1)
    sdist = torch.norm(b["src_keypts"][:, :, None, :] - b["src_keypts"][:, None, :, :], dim=-1)
    seeds_o = O.pick_seeds(sdist, lg, 0.10, S)
    _, _, fT, _, _ = O.cal_seed_trans(fn, b["src_keypts"], b["tgt_keypts"], seeds_o, float(sd["sigma"]), 0.1, 40, 10, 0.10)
    fT = O.post_refinement(fT, b["src_keypts"], b["tgt_keypts"], 0.10)
    te3 = float((res["final_trans"].cpu() - fT).abs().max())
    print(f"scene {s} N={N} T={T}: dT full {te:.2e} | HIP head on oracle logits/features {te2:.2e} (seeds equal: {np.array_equal(seeds_ref, seeds_hip)}) | oracle head on HIP logits/features {te3:.2e}")
