"""Per-kernel device time of the fused image encoder at a given batch of images (torch profiler).  GPU box:
    python tools/image_kernels_profile.py [n_images]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
enc = model.encoder._fused_image_encoder()
x = torch.rand(n, 3, 120, 160, device=dev)
with torch.no_grad():
    for _ in range(3): enc(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): enc(x)
        torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]:
    print(f"{e.key[:90]:90s} n={e.count:4d} avg={e.device_time_total / e.count:8.1f} us total={e.device_time_total / 10:8.1f} us/pass")
