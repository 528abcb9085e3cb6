"""Dev (GPU box): the prologue role launches against the six-kernel prologue - bitwise, uniform and ragged, then latency."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import _lib, synthetic
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12); model.load_state_dict(sd, strict=False); model = model.to(dev).eval()
h = _lib.handle_for(0)
KEYS = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
for B, N, T in ((1, 1000, 300), (1, 5000, 196), (2, 777, 33), (4, 3000, 300), (1, 10000, 300), (3, 200, 1), (1, 64, 700), (16, 1000, 300), (8, 1000, 196), (7, 4500, 300)):
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=T)
    d = {k: b[k].to(dev) for k in KEYS}; d["testing"] = True
    out = {}
    for knob in (0, 1):
        h.call("gmf_set_tuning", b"small_prologue_roles", knob)
        r = model(d); out[knob] = (model.last_logits.clone(), r["final_trans"].clone())
    same = torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    lens = [N - 5 * i for i in range(B)]
    rag = {k: [b[k][i, :lens[i]].to(dev) for i in range(B)] for k in KEYS[:3]}; rag.update(p_tokens=d["p_tokens"], q_tokens=d["q_tokens"], testing=True)
    ro = {}
    for knob in (0, 1):
        h.call("gmf_set_tuning", b"small_prologue_roles", knob)
        model(rag); ro[knob] = model.last_logits.clone()
    print(f"B={B} N={N} T={T}: uniform identical={same}  ragged identical={torch.equal(ro[0], ro[1])}", flush=True)
for Bt, N in ((1, 1000), (1, 5000), (1, 10000), (8, 1000), (16, 1000), (4, 5000)):
    b = synthetic.synthetic_batch(list(range(Bt)), N=N, T=300)
    d = {k: b[k].to(dev) for k in KEYS}; d["testing"] = True
    for rnd in range(3):
        row = {}
        for knob in (0, 1):
            h.call("gmf_set_tuning", b"small_prologue_roles", knob)
            for _ in range(5): model(d)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): model(d)
            torch.cuda.synchronize(); row[knob] = (time.perf_counter() - t0) / 50 * 1e3
        print(f"B={Bt} N={N}: six kernels {row[0]:.3f} ms, role launches {row[1]:.3f} ms", flush=True)
