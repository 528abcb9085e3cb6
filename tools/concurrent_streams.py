"""Serving: B = 1 forwards issued round-robin on 1 .. 4 torch streams with gmf_amd.set_handle_per_stream(True) - one library handle
(and workspace) per stream, so that the forwards overlap on the device - against the default (one handle per device: the calls are
serialised).  GPU box:  python tools/concurrent_streams.py [N ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
KEYS = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
streams = [torch.cuda.Stream() for _ in range(4)]
for N in [int(a) for a in sys.argv[1:]] or [1000, 5000]:
    datas = []
    for i in range(4):
        b = synthetic.synthetic_batch([i], N=N, T=196)
        d = {k: b[k].to(dev) for k in KEYS}; d["testing"] = True; datas.append(d)
    ref = [model(d)["final_trans"].clone() for d in datas]
    def run(ns, reps):
        out = None
        for _ in range(reps):
            out = []
            for s in range(ns):
                with torch.cuda.stream(streams[s]):
                    out.append(model(datas[s])["final_trans"])
        return out
    for per_stream in (False, True):
        gmf_amd.set_handle_per_stream(per_stream)
        for ns in (1, 2, 3, 4):
            run(ns, 5); torch.cuda.synchronize(); t0 = time.perf_counter(); out = run(ns, 50); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (50 * ns) * 1e3
            same = all(torch.equal(o, r) for o, r in zip(out, ref))
            print(f"N={N} handle per stream={per_stream}: {ns} stream(s): {dt:.3f} ms per forward = {1.0 / dt:.2f} forwards per ms; poses identical to the default stream's: {same}", flush=True)
    gmf_amd.set_handle_per_stream(False)
