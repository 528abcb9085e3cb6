import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
dev = torch.device("cuda:0")
for N0, N1, d in ((5000, 5000, 32), (8000, 8000, 32), (1000, 3000, 33), (77, 4001, 128), (5000, 5000, 32), (5000, 5000, 32)):
    a = torch.nn.functional.normalize(torch.randn(N0, d, device=dev), dim=1); b = torch.nn.functional.normalize(torch.randn(N1, d, device=dev), dim=1)
    r = gmf_amd.nn_match(a, b)
    idx = r[0] if isinstance(r, (tuple, list)) else r
    ref = torch.cdist(a.double(), b.double()).argmin(dim=1)
    for _ in range(5): gmf_amd.nn_match(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): gmf_amd.nn_match(a, b)
    torch.cuda.synchronize()
    print(N0, N1, d, "agree with fp64 argmin:", float((idx.long() == ref).float().mean()), f"{(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
