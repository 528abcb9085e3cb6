"""Data-parallel driver: scene pairs shard across the GPUs of one node, one process per GPU.

Pairs are independent in eval mode, so the data path has exactly ONE exchange step per
batch: an all-gather of one packed buffer per rank - row p = [logits of pair p (N floats) | final_trans of pair p
(16 floats)] - over RCCL/xGMI through ``torch.distributed`` backend "nccl" ("gloo" on CPU for the tests).  The
reference has no counterpart - it is single-process, single-GPU (train_3DMatch.py:17).

Uneven shards (B % world != 0) are padded to the largest shard inside the packed buffer and trimmed after the
gather, so the collective always sees equal sizes; the shard sizes come from `shard_range` (pure arithmetic, the
same on every rank - no size exchange).  The packed row carries one status float behind the pose: a rank that fails
before the collective still enters it and all ranks raise together (no rank is left waiting for a timeout).
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of pairs owned by `rank`: the first (total % world) ranks get one extra pair."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [b - a for a, b in (shard_range(total, world, r) for r in range(world))]


def shard_batch(data: Dict[str, torch.Tensor], world: int, rank: int) -> Dict[str, torch.Tensor]:
    """Slice every [B, ...] tensor of a global batch down to this rank's pairs (non-tensors pass through)."""
    B = next(v.shape[0] for v in data.values() if torch.is_tensor(v))
    lo, hi = shard_range(B, world, rank)
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in data.items()}


class ShardedBatchDriver:
    """Runs `model` on the local shard and all-gathers logits and poses with ONE collective.

    model(data) must return a dict with "final_trans" [b,4,4] and expose the logits [b,N] either as
    `model.last_logits` or as result["logits"].

    `step(local_data)` assumes every rank holds the same number of pairs (the weak-scaling benchmark) unless
    `sizes` (pairs per rank, e.g. `shard_sizes(B, world)`) is given; `run(global_data)` shards a global batch and
    passes the sizes itself.  With `always_collective=True` the exchange also runs at world size 1 when a process
    group exists (a one-rank RCCL all-gather: the same code path the multi-GPU runs take).
    """

    def __init__(self, model: Callable, world: int, rank: int, device: torch.device, backend: str = None,
                 always_collective: bool = False):
        self.model, self.world, self.rank, self.device = model, world, rank, device
        self.own_pg = False
        self.always_collective = always_collective
        self.last_model_ms: Optional[float] = None      # device time of the local forward (cuda) / wall time (cpu)
        self.last_gather_ms: Optional[float] = None     # ... of pack + all-gather + unpack
        self.time_steps = False                         # record the two figures above (adds event records, no host sync)
        # read the gathered status column after every step (one small device-to-host read): a failure on another rank then
        # raises here too instead of returning padding.  bench.py switches it off inside its timed loop and checks once after.
        self.check_status = True
        self._events = None
        # a gloo group with the model on a HIP device (a multi-process rehearsal on one GPU): collectives go through the host
        self.stage_host = False
        if world > 1 or always_collective:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL's peer buffers
                backend = backend or ("nccl" if device.type == "cuda" else "gloo")
                kw = {"device_id": device} if device.type == "cuda" else {}
                if backend == "gloo":
                    kw = {}
                dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
                self.own_pg = True
            self.stage_host = device.type == "cuda" and dist.get_backend() == "gloo"

    # -- timing helpers (no host synchronisation inside step) -------------------------------------------------
    def _mark(self, i: int):
        if not self.time_steps:
            return
        if self.device.type == "cuda":
            if self._events is None:
                self._events = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            self._events[i].record(torch.cuda.current_stream(self.device))
        else:
            if self._events is None:
                self._events = [0.0, 0.0, 0.0]
            self._events[i] = time.perf_counter()

    def read_timings(self) -> Tuple[Optional[float], Optional[float]]:
        """(model ms, gather ms) of the last step; synchronises the device - call outside timed regions."""
        if not self.time_steps or self._events is None:
            return None, None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            e = self._events
            return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
        e = self._events
        return (e[1] - e[0]) * 1e3, (e[2] - e[1]) * 1e3

    # -- the step ---------------------------------------------------------------------------------------------
    def _collective(self) -> bool:
        return (self.world > 1 or self.always_collective) and dist.is_initialized()

    @staticmethod
    def _rows_and_width(data) -> Tuple[int, int]:
        """(pairs, correspondences per pair) of a batch dict: from `corr_pos` [B, N, 6] (the key every PointDSC batch carries,
        PointDSC.py:199), else from its first [B, N, ...] tensor."""
        t = data["corr_pos"] if torch.is_tensor(data.get("corr_pos")) else \
            next(v for v in data.values() if torch.is_tensor(v) and v.dim() >= 2)
        return int(t.shape[0]), int(t.shape[1])

    def step(self, data, sizes: Optional[List[int]] = None) -> Dict[str, torch.Tensor]:
        """One sharded step.  A failure on ONE rank must not leave the others blocked in the collective: with a process
        group, a rank whose forward raises, whose data disagrees with the shard plan, or whose results do not pack, still
        enters the all-gather - with an all-padding buffer and its status column set - and EVERY rank raises after it.  The
        buffer's shape [max(sizes), N + 17] comes only from values every rank computes identically (`sizes` and N), never
        from the local batch.  A rank whose shard is empty (B < world: shard_sizes gives [1, 1, 1, 0]) skips the model and
        contributes padding only.  The one error raised BEFORE the collective is a `sizes` list of the wrong length: that is
        the caller's plan itself, identical on every rank, so every rank raises it."""
        self._mark(0)
        coll = self._collective()
        b, n = self._rows_and_width(data)
        err: Optional[BaseException] = None
        res = None
        if coll:
            if sizes is None:
                sizes = [b] * self.world
            if len(sizes) != self.world:
                raise RuntimeError(f"gmf_amd.dist: `sizes` has {len(sizes)} entries for {self.world} ranks (shard_sizes(B, world))")
            if sizes[self.rank] != b:
                err = RuntimeError(f"gmf_amd.dist: rank {self.rank} holds {b} pairs but the shard plan says {sizes}: every "
                                   "rank must pass the same `sizes` (shard_sizes(B, world)), or equal shards without it")
        bmax = max(max(sizes), 1) if coll else max(b, 1)
        logits = trans = labels = buf = None
        try:
            if err is None and b > 0:
                res = self.model(data)
            if res is not None:
                logits = res["logits"] if "logits" in res else self.model.last_logits
                trans = res["final_trans"]
                labels = res.get("final_labels")
                if coll and (tuple(logits.shape) != (b, n) or tuple(trans.shape) != (b, 4, 4)):
                    raise RuntimeError(f"gmf_amd.dist: the model returned logits {tuple(logits.shape)} / poses {tuple(trans.shape)} "
                                       f"for a shard of {b} pairs x {n} correspondences")
            if coll:
                # one packed buffer per rank: [bmax, N + 17] = logits | pose | status, rows >= b are padding
                buf = torch.zeros((bmax, n + 17), device=self.device, dtype=torch.float32)
                if res is not None:
                    buf[:b, :n] = logits
                    buf[:b, n:n + 16] = trans.reshape(b, 16)
        except Exception as e:                   # noqa: BLE001 - re-raised below, on every rank, after the collective
            if not coll:
                raise
            err, res = e, None
        if res is None:                          # empty shard, or a failed rank: padding only
            logits = torch.zeros((0, n), device=self.device)
            trans = torch.zeros((0, 4, 4), device=self.device)
            labels = None
        out = {"logits": logits, "final_trans": trans, "final_labels": labels}
        self._mark(1)
        if coll:
            if err is not None or buf is None:
                buf = torch.zeros((bmax, n + 17), device=self.device, dtype=torch.float32)
                buf[:, n + 16] = 1.0
            if self.stage_host:
                g_host = torch.empty((self.world * bmax, n + 17), dtype=torch.float32)
                dist.all_gather_into_tensor(g_host, buf.cpu())
                gathered = g_host.to(self.device)
            else:
                gathered = torch.empty((self.world * bmax, n + 17), device=self.device, dtype=torch.float32)
                dist.all_gather_into_tensor(gathered, buf)       # the ONE exchange step of the batch
            failed = [r for r in range(self.world) if float(gathered[r * bmax, n + 16]) != 0.0] if (err is not None or self.check_status) else []
            if failed:
                raise RuntimeError(f"gmf_amd.dist: the step failed on rank(s) {failed}; every rank leaves the collective and "
                                   f"raises (rank {self.rank}: {err if err is not None else 'ok'})") from err
            if min(sizes) != bmax:
                gathered = torch.cat([gathered[r * bmax:r * bmax + sizes[r]] for r in range(self.world)])
            out["all_logits"] = gathered[:, :n]
            out["all_trans"] = gathered[:, n:n + 16].reshape(-1, 4, 4)
        else:
            out["all_logits"], out["all_trans"] = logits, trans
        self._mark(2)
        return out

    def run(self, global_data) -> Dict[str, torch.Tensor]:
        """Shard a global batch [B, ...] over the ranks (uneven allowed), run, gather."""
        B = next(v.shape[0] for v in global_data.values() if torch.is_tensor(v))
        return self.step(shard_batch(global_data, self.world, self.rank), sizes=shard_sizes(B, self.world))

    def barrier(self):
        if self.world > 1 and dist.is_initialized():
            dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.stage_host else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, value: float) -> List[float]:
        """One float from every rank (diagnostics: per-rank step times)."""
        if self.world == 1:
            return [value]
        dev = "cpu" if self.stage_host else self.device
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        allv = torch.empty(self.world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allv, t)
        return [float(v) for v in allv.cpu()]

    def gather_objects(self, obj) -> list:
        """One picklable object from every rank (diagnostics: which device each rank ran on)."""
        if self.world == 1 or not dist.is_initialized():
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def group_size(self) -> int:
        """Ranks of the process group as the backend reports them (1 without a group)."""
        return dist.get_world_size() if dist.is_initialized() else 1

    def backend_name(self) -> str:
        return str(dist.get_backend()) if dist.is_initialized() else "none"

    def close(self):
        if self.own_pg and dist.is_initialized():
            dist.destroy_process_group()
