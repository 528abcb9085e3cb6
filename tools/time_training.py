"""One training step (forward + losses + backward + Adam) of gmf_amd.PointDSC in train() mode at the reference's training
configuration (config_3DMatch.py: batch 16, 1000 correspondences, 12 layers; 300 image tokens), against the same step with
the image tokens given.  GPU box:  python tools/time_training.py [B] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
m = gmf_amd.PointDSC(num_layers=12); m.load_state_dict(sd, strict=False); m = m.to(dev).train()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=300)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
gt = b["gt_labels"].to(dev)
cl_fn, sm_fn = gmf_amd.ClassificationLoss(balanced=False), gmf_amd.SpectralMatchingLoss(balanced=False)
FUSED = os.environ.get("ADAM", "fused") == "fused"      # ADAM=foreach: torch's default (multi-tensor) implementation
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6, fused=FUSED)
def step():
    opt.zero_grad()
    res = m(data)
    loss = cl_fn(res["final_labels"], gt)["loss"] + sm_fn(res["M"], gt)
    loss.backward()
    opt.step()
    return float(loss.detach())
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"training step B={B} N={N} T=300, 12 layers, Adam {'fused' if FUSED else 'foreach'}: {dt * 1e3:.1f} ms  (loss {l:.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB + workspace)")
# [r5] the same step captured as a HIP graph (gmf_amd.train.GraphedTrainingStep): sigma on the device, loss statistics on the device, capturable Adam
from gmf_amd import train as T
m2 = gmf_amd.PointDSC(num_layers=12); m2.load_state_dict(sd, strict=False); m2 = m2.to(dev).train()
cl2 = gmf_amd.ClassificationLoss(balanced=False, host_stats=False)
opt2 = torch.optim.Adam([p for n, p in m2.named_parameters() if not n.startswith("encoder.image_encoder.")], lr=1e-4, weight_decay=1e-6, capturable=True, fused=FUSED)
data2 = dict(data, gt=gt)
gstep = T.GraphedTrainingStep(m2, opt2, lambda res, bt: cl2(res["final_labels"], bt["gt"])["loss"] + sm_fn(res["M"], bt["gt"]), data2, warmup=3)
for _ in range(2): gstep(data2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): lg = gstep(data2)
torch.cuda.synchronize()
print(f"the same step as ONE captured HIP graph: {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms  (loss {float(lg.detach()):.4f})")
def fwd():
    with torch.no_grad():
        m.eval(); m(data); m.train()
for _ in range(2): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fwd()
torch.cuda.synchronize()
print(f"eval-mode forward (fused inference kernels) for comparison: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
if os.environ.get("PROFILE", "0") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]
    for e in rows:
        print(f"{e.key[:90]:90s} n={e.count:5d} avg={e.device_time_total / e.count:9.1f} us total={e.device_time_total / 1e3:8.2f} ms")
