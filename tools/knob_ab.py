"""Time PointDSC.forward under settings of one tuning knob.  GPU box:  python tools/knob_ab.py B N KNOB v1,v2,..."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import _lib, synthetic
B, N, knob = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3].encode()
vals = [int(v) for v in sys.argv[4].split(",")]
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
h = _lib.handle_for(0)
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
data["testing"] = True
for rnd in range(2):
    for v in vals:
        h.call("gmf_set_tuning", knob, v)
        for _ in range(3): model(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(data)
        torch.cuda.synchronize()
        if rnd: print(f"B={B} N={N} {knob.decode()}={v}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
