"""Dev (GPU box): which Python frames issue a given aten op during one training step.
    python tools/who_launches.py aten::div [B] [N]"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
op = sys.argv[1] if len(sys.argv) > 1 else "aten::div"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
m = gmf_amd.PointDSC(num_layers=12); m.load_state_dict(sd, strict=False); m = m.to(dev).train()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=300)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
gt = b["gt_labels"].to(dev)
cl_fn, sm_fn = gmf_amd.ClassificationLoss(balanced=False), gmf_amd.SpectralMatchingLoss(balanced=False)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
def step():
    opt.zero_grad()
    res = m(data)
    loss = cl_fn(res["final_labels"], gt)["loss"] + sm_fn(res["M"], gt)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ops = collections.Counter(); where = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"):
        ops[e.name] += 1
        if e.name.startswith(op):
            st = [f for f in (e.stack or []) if "site-packages/torch" not in f][:2]
            where[" <- ".join(st) or "(no python frame: autograd thread)"] += 1
print("aten ops of one step:", ops.most_common(80))
print(f"{op}:")
for k, v in where.most_common(12): print(f"  {v:5d}  {k}")
