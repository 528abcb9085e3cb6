"""Image encoder (64 images of 120 x 160 = the 32-pair step) with the native layer1 / layer2 convolutions against MIOpen's,
alternating in one process; eager and as a captured HIP graph.  GPU box:  python tools/image_encoder_ab.py [n_images]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
enc = model.encoder._fused_image_encoder()
x = torch.rand(n, 3, 120, 160, device=dev).contiguous(memory_format=torch.channels_last)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    ref = None
    for rnd in range(2):
        for name, thr in (("MIOpen convolutions", 1 << 60), ("native layer1/2    ", 0)):
            enc.min_native_pixels = thr
            y = enc(x)
            if ref is None: ref = y
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = enc(x)
            print(f"{name}: eager {t(lambda: enc(x)):.3f} ms, graph {t(lambda: g.replay()):.3f} ms, max |diff| vs MIOpen {float((y - ref).abs().max()):.2e}")
