import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import gmf_amd
from gmf_amd import _lib, synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
h = _lib.handle_for(0)
for B, N in ((32, 1000), (36, 1000), (40, 1000), (48, 1000), (64, 1000), (7, 5000), (8, 5000), (10, 5000), (12, 5000), (4, 10000), (32, 5000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}; data["testing"] = True
    row = []
    ref = None
    for roles in (0, 100000):
        h.call("gmf_set_tuning", b"mid_grid_roles", min(roles, 4096))
        for _ in range(3): model(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): model(data)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 20 * 1e3)
        lg = model.last_logits.clone()
        if ref is None: ref = lg
        else: same = bool(torch.equal(ref, lg))
    W = ((N + 31) // 32 + 3) // 4 * B
    print(f"B={B} N={N} W={W}: one kernel {row[0]:.3f} ms | two roles {row[1]:.3f} ms  identical={same}", flush=True)
