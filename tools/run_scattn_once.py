"""Runs the encoder a few times with one attention variant (for rocprofv3 --pmc passes).  GPU box only:
    python tools/run_scattn_once.py VARIANT [B] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import _lib, synthetic              # noqa: E402

v = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
args = [b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
_lib.handle_for(0).call("gmf_set_tuning", b"scattn_variant", v)
if os.environ.get("GMF_PRECISION"):                  # the throughput numerics mode (tools only: the library reads no environment)
    _lib.handle_for(0).call("gmf_set_tuning", b"precision", int(os.environ["GMF_PRECISION"]))
for _ in range(2):
    model.encode(*args)
torch.cuda.synchronize()
