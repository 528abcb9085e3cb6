import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gmf_amd
from gmf_amd import synthetic
g = np.load("/root/repo/tests/golden/f13_global_registration.npz")
for c in g["cases"]:
    N, seed, ratio, q, use_w = int(c[0]), int(c[1]), float(c[2]), float(c[3]), bool(c[4])
    X, Y, w, _, _ = synthetic.dgr_scene(N, seed)
    R, t, o = gmf_amd.GlobalRegistration(X.cuda(), Y.cuda(), weights=w.cuda() if use_w else None, break_threshold_ratio=ratio, quantization_size=q)
    tag = f"{N}_{seed}"
    print(tag, o, g[f"stats_{tag}"], "dR %.2e dt %.2e" % (np.abs(R.cpu().numpy() - g[f"R_{tag}"]).max(), np.abs(t.cpu().numpy() - g[f"t_{tag}"]).max()))
import time
X, Y, w, _, _ = synthetic.dgr_scene(8000, 2)
X, Y, w = X.cuda(), Y.cuda(), w.cuda()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): gmf_amd.GlobalRegistration(X, Y, weights=w, break_threshold_ratio=1e-4, quantization_size=0.1)
torch.cuda.synchronize(); print("ms per solve (N=8000):", (time.perf_counter() - t0) * 100)
