"""Per-kernel device time of one encoder pass (GPU box): wall time of whole passes + a torch-profiler table of the kernels.
    python tools/kernel_times.py [B] [N] [reps] [knob=value ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import _lib, synthetic              # noqa: E402

if os.environ.get("GMF_LIB"):                    # tools/ubench/ablate_h2p.py: time an ablated (wrong-result) build
    _lib.LIB_PATH = os.environ["GMF_LIB"]
    _lib.Handle.raise_if_flagged = lambda self, where: self.status(clear=True)   # (its NaNs are expected: timing only)

pos = [a for a in sys.argv[1:] if "=" not in a]
knobs = [a.split("=") for a in sys.argv[1:] if "=" in a]
B = int(pos[0]) if len(pos) > 0 else 32
N = int(pos[1]) if len(pos) > 1 else 5000
reps = int(pos[2]) if len(pos) > 2 else 5
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
for k, v in knobs:
    _lib.handle_for(0).call("gmf_set_tuning", k.encode(), int(v))
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
args = [b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
data["testing"] = True
if os.environ.get("FULL", "0") == "1":          # whole forward (encoder + pose head) instead of the encoder alone
    class _M:
        @staticmethod
        def encode(*a):
            return model(data)
    run = _M
else:
    run = model
for _ in range(2):
    run.encode(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run.encode(*args)
torch.cuda.synchronize()
print(f"encode: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per pass (B={B}, N={N}, knobs={knobs})")
from torch.profiler import profile, ProfilerActivity   # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(reps):
        run.encode(*args)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:int(os.environ.get('ROWS', '12'))]
for e in rows:
    print(f"{e.key[:70]:70s} n={e.count:4d} avg={e.device_time_total / e.count:9.1f} us total={e.device_time_total / reps / 1e3:7.3f} ms/pass")
