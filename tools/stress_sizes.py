import sys, time, torch
sys.path.insert(0, "/root/repo")
import gmf_amd
from gmf_amd import synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
for B, N in ((64, 5000), (8, 16000), (256, 1000), (2, 16384)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    out = model(data); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = model(data); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    T = out["final_trans"].cpu(); gt = b["gt_trans"]
    errR = float((T[:, :3, :3] - gt[:, :3, :3]).abs().max())
    print(f"B={B} N={N}: {dt*1e3:.1f} ms, {B*N/dt/1e6:.2f} M corr/s, finite={bool(torch.isfinite(T).all())}, max R err vs gt {errR:.3f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB torch")
