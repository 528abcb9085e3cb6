"""Per-kernel device time of one whole PointDSC.forward (test mode) from raw images or tokens (torch profiler).  GPU box:
    python tools/forward_kernels_profile.py [B] [N] [images|tokens] [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
src = sys.argv[3] if len(sys.argv) > 3 else "images"
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=300)
d = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
d["testing"] = True
if src == "images":
    d.update(p_image=torch.rand(B, 3, 120, 160, device=dev), q_image=torch.rand(B, 3, 120, 160, device=dev))
else:
    d.update(p_tokens=b["p_tokens"].to(dev), q_tokens=b["q_tokens"].to(dev))
import time
for _ in range(3): model(d)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): model(d)
torch.cuda.synchronize()
print(f"B={B} N={N} from {src}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): model(d)
    torch.cuda.synchronize()
tot = 0.0
evs = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
for e in evs:
    tot += e.device_time_total / 10
print(f"device time of all kernels: {tot:.1f} us per forward, {sum(e.count for e in evs) / 10:.0f} launches")
for e in evs[:rows]:
    print(f"{e.key[:100]:100s} n={e.count / 10:5.1f} avg={e.device_time_total / e.count:8.1f} us total={e.device_time_total / 10:8.1f} us")
