"""One-rank RCCL sanity check on the GPU box: the collectives bench.py --gpus N uses (all_gather_into_tensor, all_reduce MAX,
barrier) on backend "nccl" with world size 1.  python tools/rccl_sanity.py"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(32 * 5000, dtype=torch.float32, device=dev).reshape(32, 5000)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, x) and float(t) == 1.5
print("RCCL one-rank collectives OK:", torch.cuda.nccl.version())
dist.destroy_process_group()
