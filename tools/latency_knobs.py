"""Dev A/B (GPU box): B = 1 forward latency under tuning knobs, interleaved in one process.
    python tools/latency_knobs.py N knob=v1,v2,... [knob2=...]"""
import itertools, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import _lib, synthetic
N = int(sys.argv[1])
B = int(os.environ.get("PAIRS", "1"))
knobs = {a.split("=")[0]: [int(v) for v in a.split("=")[1].split(",")] for a in sys.argv[2:]}
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}; data["testing"] = True
h = _lib.handle_for(0)
names = list(knobs)
for rnd in range(3):
    row = {}
    for combo in itertools.product(*[knobs[n] for n in names]):
        for n, v in zip(names, combo): h.call("gmf_set_tuning", n.encode(), v)
        for _ in range(3): model(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): model(data)
        torch.cuda.synchronize()
        row[combo] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    print(rnd, names, row)
