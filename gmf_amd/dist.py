"""Data-parallel driver: scene pairs shard across the GPUs of one node, one process per GPU.

Pairs are independent in eval mode, so the data path has exactly one exchange step:
an all-gather of the per-pair inlier logits [B/G, N] and poses [B/G, 4, 4] (RCCL over xGMI through
``torch.distributed`` backend "nccl"; "gloo" on CPU for the tests).  The reference has no counterpart - it
is single-process, single-GPU (train_3DMatch.py:17).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of pairs owned by `rank`: the first (total % world) ranks get one extra pair."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(data: Dict[str, torch.Tensor], world: int, rank: int) -> Dict[str, torch.Tensor]:
    """Slice every [B, ...] tensor of a global batch down to this rank's pairs (non-tensors pass through)."""
    B = next(v.shape[0] for v in data.values() if torch.is_tensor(v))
    lo, hi = shard_range(B, world, rank)
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in data.items()}


class ShardedBatchDriver:
    """Runs `model` on the local shard and all-gathers logits and poses.

    model(data) must return a dict with "final_trans" [b,4,4] and expose the logits [b,N] either as
    `model.last_logits` or as result["logits"].
    """

    def __init__(self, model: Callable, world: int, rank: int, device: torch.device, backend: str = None):
        self.model, self.world, self.rank, self.device = model, world, rank, device
        self.own_pg = False
        if world > 1:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                backend = backend or ("nccl" if device.type == "cuda" else "gloo")
                kw = {"device_id": device} if device.type == "cuda" else {}
                dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
                self.own_pg = True

    def step(self, data) -> Dict[str, torch.Tensor]:
        res = self.model(data)
        logits = res["logits"] if "logits" in res else self.model.last_logits
        out = {"logits": logits, "final_trans": res["final_trans"], "final_labels": res.get("final_labels")}
        if self.world > 1:
            b, n = logits.shape
            all_logits = torch.empty((self.world * b, n), device=logits.device, dtype=logits.dtype)
            all_trans = torch.empty((self.world * b, 4, 4), device=logits.device, dtype=logits.dtype)
            dist.all_gather_into_tensor(all_logits, logits.contiguous())
            dist.all_gather_into_tensor(all_trans, res["final_trans"].contiguous())
            out["all_logits"], out["all_trans"] = all_logits, all_trans
        else:
            out["all_logits"], out["all_trans"] = logits, res["final_trans"]
        return out

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.own_pg and dist.is_initialized():
            dist.destroy_process_group()
