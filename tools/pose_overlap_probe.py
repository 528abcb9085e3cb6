"""Probe: how much of the pose head (latency-bound small kernels at the tail of a step) hides under the NEXT step's encoder when
it runs on a side stream with a library handle (workspace) of its own.  GPU box:  python tools/pose_overlap_probe.py [B] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import _lib, synthetic, pointdsc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
data = {k: b[k].to(dev) for k in keys}
data["testing"] = True
args = [data[k] for k in keys]
side = torch.cuda.Stream(device=dev)
hB = _lib.Handle(0)
orig = pointdsc.handle_and_stream


def step_serial():
    return model(data)


def step_overlap():
    logits, feat_n, _ = model.encode(*args)
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        pointdsc.handle_and_stream = lambda t, check=False: (hB, side.cuda_stream)
        try:
            T, lab, _ = model.pose_head(feat_n, data["src_keypts"], data["tgt_keypts"], logits, True)
        finally:
            pointdsc.handle_and_stream = orig
    logits.record_stream(side); feat_n.record_stream(side)
    return T, lab


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): out = fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best, out


ms_s, out_s = bench(step_serial)
ms_o, out_o = bench(step_overlap)
ms_e, _ = bench(lambda: model.encode(*args))
torch.cuda.synchronize()
print(f"B={B} N={N}: serial step {ms_s:.3f} ms, encoder alone {ms_e:.3f} ms, pose head on a side stream {ms_o:.3f} ms per step; "
      f"same pose: {bool(torch.equal(out_s['final_trans'], out_o[0]))}")
