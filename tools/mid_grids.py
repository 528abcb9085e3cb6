import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import gmf_amd
from gmf_amd import _lib
if os.environ.get("GMF_LIB"): _lib.LIB_PATH = os.environ["GMF_LIB"]
from gmf_amd import synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
for B, N in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((32, 1000), (40, 1000), (48, 1000), (64, 1000), (8, 5000), (12, 5000), (4, 10000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}; data["testing"] = True
    for _ in range(3): model(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): model(data)
    torch.cuda.synchronize()
    W = ((N + 31) // 32 + 3) // 4 * B
    print(f"B={B} N={N} W={W}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
