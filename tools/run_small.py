"""One small-batch configuration a few times (for rocprofv3 kernel traces): python tools/run_small.py B N"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd                                   # noqa: E402
from gmf_amd import synthetic                    # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
model = gmf_amd.PointDSC(num_layers=12)
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
data["testing"] = True
for _ in range(5):
    model(data)
torch.cuda.synchronize()
