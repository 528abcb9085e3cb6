"""Measure the 16-bit compat-cache formats (gmf_set_tuning "compat_format" = 2: 16-bit fixed point, 3: fp16 of 1 - c) against the
fp32 cache, in the PARITY kernel: time per step and per attention launch at the headline size, the logit / pose deviation
from the fp32-cache result there, and the randomised parity sweep (HIP vs the fp32 oracle and vs an fp64 evaluation) for
every format on the same scenes.

GPU box:  python tests/tools/compat_format_ab.py [n_sweep_scenes] [B] [N]"""
import ctypes as C
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import _lib, synthetic
from oracle import gmf_oracle as O

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
FMTS = [0, 2, 3, 4]
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
h = _lib.handle_for(0)


def to_dev(b):
    d = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    d["testing"] = True
    return d


# ---- timing + deviation at the headline size -------------------------------------------------------------------------------
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
data = to_dev(b)
ref_logits = ref_T = None
for rnd in range(2):
    for f in FMTS:
        h.call("gmf_set_tuning", b"compat_format", f)
        for _ in range(3): res = model(data)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10): res = model(data)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        if rnd == 0: continue
        h.call("gmf_profile_enable", 1)
        for _ in range(5): model(data)
        torch.cuda.synchronize()
        ms, n = C.c_double(), C.c_int()
        h.call("gmf_profile_read", C.byref(ms), C.byref(n))
        h.call("gmf_profile_enable", 0)
        lg, T, lab = model.last_logits.clone(), res["final_trans"].clone(), res["final_labels"].clone()
        if f == 0: ref_logits, ref_T, ref_lab = lg, T, lab
        print(f"B={B} N={N} compat_format={f}: step {best * 1e3:.3f} ms, attention {ms.value / max(n.value, 1):.4f} ms per launch "
              f"({n.value} launches); vs fp32 cache: max|dlogit| {float((lg - ref_logits).abs().max()):.3e} "
              f"mean {float((lg - ref_logits).abs().mean()):.3e}, max|dT| {float((T - ref_T).abs().max()):.3e}, "
              f"labels differing {int((lab != ref_lab).sum())}", flush=True)

# ---- parity sweep: every format on the same scenes ---------------------------------------------------------------------------
rng = np.random.default_rng(2024)
torch.set_num_threads(16)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
err = {f: [] for f in FMTS}; err64 = {f: [] for f in FMTS}; terr = {f: [] for f in FMTS}; ref64 = []
t0 = time.time()
for s in range(n_scenes):
    Ns = int(rng.choice([64, 200, 333, 500, 777, 1000, 1500, 2048, 3000]))
    T = int(rng.choice([40, 196, 300]))
    bb = synthetic.synthetic_batch([1000 + s], N=Ns, T=T)
    ref = O.pointdsc_forward(sd, bb, testing=True)
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in bb.items()}
    compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
    truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))
    ref64.append(float((ref["logits"].double() - truth).abs().max()))
    dd = to_dev(bb)
    for f in FMTS:
        h.call("gmf_set_tuning", b"compat_format", f)
        res = model(dd)
        lg = model.last_logits.cpu()
        err[f].append(float((lg - ref["logits"]).abs().max()))
        err64[f].append(float((lg.double() - truth).abs().max()))
        terr[f].append(float((res["final_trans"].cpu() - ref["final_trans"]).abs().max()))
    if (s + 1) % 10 == 0:
        print(f"{s + 1} scenes, {time.time() - t0:.0f} s: " + "  ".join(f"fmt{f} max {max(err[f]):.2e}" for f in FMTS), flush=True)
r = np.array(ref64)
print(f"fp32 oracle vs fp64: median {np.median(r):.2e} p90 {np.quantile(r, .9):.2e} max {r.max():.2e}")
for f in FMTS:
    e, e6, te = np.array(err[f]), np.array(err64[f]), np.array(terr[f])
    print(f"compat_format={f}: vs fp32 oracle median {np.median(e):.2e} p90 {np.quantile(e, .9):.2e} max {e.max():.2e} (above 1e-4: {(e > 1e-4).sum()} of {len(e)})"
          f"  |  vs fp64 median {np.median(e6):.2e} p90 {np.quantile(e6, .9):.2e} max {e6.max():.2e}  |  above oracle-vs-fp64 + 2e-5: {(e6 > r + 2e-5).sum()}"
          f"  |  pose max {te.max():.2e}")

# ---- KITTI shape (sigma_d = 1.2, coordinates of +-40 m: large attention logits, the hard case for a quantised c) --------------
sdk = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2)
mk = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=1.2, sigma_d=1.2,
                      k=40, nms_radius=1.2)
mk.load_state_dict(sdk, strict=False); mk = mk.to(dev).eval()
sdk64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sdk.items()}
for seed, Nk, Tk in ((83, 700, 50), (84, 1500, 196), (85, 3000, 196), (86, 5000, 196)):
    bb = synthetic.synthetic_batch([seed], N=Nk, T=Tk, kind="kitti")
    ref = O.pointdsc_forward(sdk, bb, inlier_threshold=1.2, nms_radius=1.2, testing=True)
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in bb.items()}
    compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 1.2)
    truth = O.classifier(sdk64, O.encoder(sdk64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))
    floor = float((ref["logits"].double() - truth).abs().max())
    dd = to_dev(bb)
    row = []
    for f in FMTS:
        h.call("gmf_set_tuning", b"compat_format", f)
        res = mk(dd)
        lg = mk.last_logits.cpu()
        row.append(f"fmt{f}: vs oracle {float((lg - ref['logits']).abs().max()):.2e} vs fp64 {float((lg.double() - truth).abs().max()):.2e} "
                   f"dT {float((res['final_trans'].cpu() - ref['final_trans']).abs().max()):.1e}")
    print(f"kitti N={Nk} T={Tk}: fp32 oracle vs fp64 {floor:.2e} | " + " | ".join(row), flush=True)
h.call("gmf_set_tuning", b"compat_format", 2)
