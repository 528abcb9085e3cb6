"""Cost of the image encoder (ResNet-34 -> layer2, PyTorch-ROCm / MIOpen) in front of the hot path:
PointDSC.forward from raw p_image / q_image [B,3,120,160] against the token path.  GPU box only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
for B, N in ((1, 1000), (32, 5000)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=300)
    base = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    base["testing"] = True
    img = {"p_image": torch.rand(B, 3, 120, 160, device=dev), "q_image": torch.rand(B, 3, 120, 160, device=dev)}
    tok = {"p_tokens": b["p_tokens"].to(dev), "q_tokens": b["q_tokens"].to(dev)}
    for name, extra in (("tokens", tok), ("images", img)):
        d = dict(base, **extra)
        for _ in range(3): model(d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(d)
        torch.cuda.synchronize()
        print(f"B={B} N={N} from {name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
    enc = model.encoder
    for _ in range(3): enc.image_tokens(img["p_image"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): enc.image_tokens(img["p_image"]); enc.image_tokens(img["q_image"])
    torch.cuda.synchronize()
    print(f"B={B}: image encoder alone (both images): {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
