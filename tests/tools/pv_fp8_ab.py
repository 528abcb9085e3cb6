"""A/B of the attention kernel's "pv_fp8" form (gmf_set_tuning "pv_fp8" = 1: the two cross products of O += P V on the block-scaled
fp8 matrix pipe; 0: all three products on the f16 pipe): time per step and per attention launch at the headline size, the
deviation between the two forms there, and - because the form only runs on LARGE grids - a parity sweep in which the scenes
travel as ragged batches (ragged batches always take the large-grid path): every scene against the fp32 oracle and against
an fp64 evaluation, 3DMatch shape and KITTI shape.

GPU box:  python tests/tools/pv_fp8_ab.py [n_sweep_batches] [B] [N]"""
import ctypes as C
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gmf_amd
from gmf_amd import _lib, synthetic
from oracle import gmf_oracle as O

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
FORMS = [0, 1, 2]      # three f16 products | fp8 cross products under the device-side guard (default) | unconditionally
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
h = _lib.handle_for(0)
torch.set_num_threads(16)


def to_dev(b):
    d = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    d["testing"] = True
    return d


# ---- timing + deviation at the headline size -------------------------------------------------------------------------------
if B > 0:
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = to_dev(b)
    ref_logits = None
    for rnd in range(2):
        for f in FORMS:
            h.call("gmf_set_tuning", b"pv_fp8", f)
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
            print(f"B={B} N={N} pv_fp8={f}: step {best * 1e3:.3f} ms, attention {ms.value / max(n.value, 1):.4f} ms per launch "
                  f"({n.value} launches); vs pv_fp8=0: max|dlogit| {float((lg - ref_logits).abs().max()):.3e} "
                  f"mean {float((lg - ref_logits).abs().mean()):.3e}, max|dT| {float((T - ref_T).abs().max()):.3e}, "
                  f"labels differing {int((lab != ref_lab).sum())}", flush=True)


# ---- parity sweeps through ragged batches (the large-grid path at any N) ----------------------------------------------------
def sweep(kind, sigma_d, mdl, sdict, sizes, seeds_of_batch, n_b):
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sdict.items()}
    err = {f: [] for f in FORMS}; err64 = {f: [] for f in FORMS}; terr = {f: [] for f in FORMS}; ref64 = []
    rng = np.random.default_rng(77)
    t0 = time.time()
    for bi in range(n_b):
        seeds = seeds_of_batch(bi)
        scenes = [synthetic.synthetic_batch([s], N=int(rng.choice(sizes)), T=196, kind=kind) for s in seeds]
        refs, truths = [], []
        for bb in scenes:
            kw = {} if kind == "3dmatch" else {"inlier_threshold": sigma_d, "nms_radius": sigma_d}
            refs.append(O.pointdsc_forward(sdict, bb, testing=True, **kw))
            b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in bb.items()}
            compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], sigma_d)
            truths.append(O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))[0])
            ref64.append(float((refs[-1]["logits"][0].double() - truths[-1]).abs().max()))
        data = {"corr_pos": [bb["corr_pos"][0].to(dev) for bb in scenes], "src_keypts": [bb["src_keypts"][0].to(dev) for bb in scenes],
                "tgt_keypts": [bb["tgt_keypts"][0].to(dev) for bb in scenes],
                "p_tokens": torch.cat([bb["p_tokens"] for bb in scenes]).to(dev), "q_tokens": torch.cat([bb["q_tokens"] for bb in scenes]).to(dev)}
        for f in FORMS:
            h.call("gmf_set_tuning", b"pv_fp8", f)
            res = mdl.forward_ragged(data)
            for i, bb in enumerate(scenes):
                lg = res["logits"][i].cpu()
                err[f].append(float((lg - refs[i]["logits"][0]).abs().max()))
                err64[f].append(float((lg.double() - truths[i]).abs().max()))
                terr[f].append(float((res["final_trans"][i].cpu() - refs[i]["final_trans"][0]).abs().max()))
        print(f"{kind}: {bi + 1} batches, {time.time() - t0:.0f} s: " + "  ".join(f"pv_fp8={f} max vs oracle {max(err[f]):.2e}" for f in FORMS), flush=True)
    r = np.array(ref64)
    print(f"{kind}: fp32 oracle vs fp64: median {np.median(r):.2e} p90 {np.quantile(r, .9):.2e} max {r.max():.2e}  ({len(r)} scenes)")
    for f in FORMS:
        e, e6, te = np.array(err[f]), np.array(err64[f]), np.array(terr[f])
        print(f"{kind} pv_fp8={f}: vs fp32 oracle median {np.median(e):.2e} p90 {np.quantile(e, .9):.2e} max {e.max():.2e} (above 1e-4: {(e > 1e-4).sum()} of {len(e)})"
              f"  |  vs fp64 median {np.median(e6):.2e} p90 {np.quantile(e6, .9):.2e} max {e6.max():.2e}  |  above 1.5 x (oracle vs fp64) + 2e-5: {(e6 > 1.5 * r + 2e-5).sum()}"
              f"  |  pose median {np.median(te):.2e} max {te.max():.2e}", flush=True)


sweep("3dmatch", 0.1, model, sd, [64, 200, 333, 500, 777, 1000, 1500, 2048, 3000], lambda bi: [1000 + 8 * bi + i for i in range(8)], n_batches)
sdk = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2)
mk = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=1.2, sigma_d=1.2,
                      k=40, nms_radius=1.2)
mk.load_state_dict(sdk, strict=False); mk = mk.to(dev).eval()
sweep("kitti", 1.2, mk, sdk, [700, 1500, 3000, 5000], lambda bi: [83 + 4 * bi + i for i in range(4)], max(1, n_batches // 2))
h.call("gmf_set_tuning", b"pv_fp8", 1)
