"""N>1 path on CPU: world_size-2 gloo ranks run the sharded driver with a stand-in per-pair function and must
reproduce the single-process result bit for bit (shard-vs-single equivalence)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_model(data):
    """Deterministic per-pair function of the inputs only (pairs never interact, like the real model)."""
    x = data["corr_pos"]
    logits = (x * torch.arange(1, 7, dtype=x.dtype)).sum(-1).sin()
    T = torch.eye(4).repeat(x.shape[0], 1, 1)
    T[:, :3, 3] = data["src_keypts"].mean(1)
    return {"logits": logits, "final_trans": T, "final_labels": (logits > 0).float()}


def _global_batch(B=5, N=37):
    g = torch.Generator().manual_seed(1234)
    return {"corr_pos": torch.randn(B, N, 6, generator=g), "src_keypts": torch.randn(B, N, 3, generator=g), "testing": True}


def _worker(rank, world, port, q, B):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gmf_amd.dist import ShardedBatchDriver, shard_batch
    drv = ShardedBatchDriver(_fake_model, world, rank, torch.device("cpu"), backend="gloo")
    drv.time_steps = True
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    if B % world == 0:
        out = drv.step(shard_batch(_global_batch(B=B), world, rank))      # equal shards: local data, no plan needed
    else:
        out = drv.run(_global_batch(B=B))                                   # uneven shards: padded inside the one buffer
    n_collectives = len(calls)
    ms = drv.read_timings()
    bad = None
    if B % world != 0:
        try:                                                                # local data of unequal size WITHOUT a plan:
            drv.step(shard_batch(_global_batch(B=B), world, rank), sizes=[1] * world)   # must raise, not hang
        except RuntimeError as e:
            bad = str(e)
    dist.all_gather_into_tensor = real
    drv.barrier()
    t = drv.max_over_ranks(float(rank + 1))
    per_rank = drv.gather_floats(float(10 + rank))
    q.put((rank, out["all_logits"].clone(), out["all_trans"].clone(), t, n_collectives, bad, per_rank, ms))
    drv.close()


def test_shard_range_covers_batch():
    from gmf_amd.dist import shard_range
    for total in (1, 5, 8, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


import pytest


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _failing_worker(rank, world, port, q):
    """Rank 1's forward raises before the collective: rank 0 must not be left waiting in the all-gather - both ranks raise."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from gmf_amd.dist import ShardedBatchDriver

    def model(data):
        if rank == 1:
            raise ValueError("boom on rank 1")
        return _fake_model(data)

    drv = ShardedBatchDriver(model, world, rank, torch.device("cpu"), backend="gloo")
    msg = None
    try:
        drv.run(_global_batch(B=4))
    except RuntimeError as e:
        msg = str(e)
    # the group is still usable afterwards (nobody is stuck inside a collective)
    out = ShardedBatchDriver(_fake_model, world, rank, torch.device("cpu")).run(_global_batch(B=4))
    q.put((rank, msg, out["all_logits"].clone()))
    drv.close()


def test_failure_on_one_rank_raises_on_every_rank():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_model(_global_batch(B=4))
    for rank, msg, logits in got:
        assert msg is not None and "rank(s) [1]" in msg, msg
        assert ("boom" in msg) == (rank == 1)
        assert torch.equal(logits, ref["logits"])


@pytest.mark.parametrize("B", [6, 5, 1])
def test_two_rank_gloo_matches_single_process(B):
    """6 pairs over 2 ranks (equal shards), 5 pairs over 2 ranks (3 + 2: padded to the larger shard inside the packed
    buffer and trimmed after the gather) and ONE pair over 2 ranks (rank 1's shard is empty: it skips the model and
    contributes padding only): every rank ends with the single-process result bit for bit, after exactly ONE
    collective per step."""
    world = 2
    port = _free_port()                  # (a fixed port can still be held by a previous run)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, B)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_model(_global_batch(B=B))
    for rank, all_logits, all_trans, t, n_coll, bad, per_rank, ms in got:
        assert torch.equal(all_logits, ref["logits"])
        assert torch.equal(all_trans, ref["final_trans"])
        assert t == 2.0          # MAX over ranks of (rank + 1)
        assert n_coll == 1       # one all_gather_into_tensor per step (logits and poses packed together)
        assert per_rank == [10.0, 11.0]
        assert ms[0] is not None and ms[0] >= 0 and ms[1] >= 0
        if B % world != 0:
            # every rank raises after the collective; the rank(s) whose data disagrees with the plan say why
            assert bad is not None and "failed on rank(s)" in bad
            assert ("shard plan" in bad) == (rank == 1 or B == 5)


def _plan_worker(rank, world, port, q, B, N):
    """One rank of BASELINE config 4's plan (B pairs over 8 ranks): the sharded step on its shard, one collective, the gathered
    result back through the queue as checksums (rank 0 also sends the tensors)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import torch.distributed as dist
    from gmf_amd.dist import ShardedBatchDriver, shard_sizes
    drv = ShardedBatchDriver(_fake_model, world, rank, torch.device("cpu"), backend="gloo")
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    out = drv.run(_global_batch(B=B, N=N))
    dist.all_gather_into_tensor = real
    local = int(out["logits"].shape[0])
    t = drv.max_over_ranks(float(rank + 1))
    q.put((rank, local, len(calls), t, float(out["all_logits"].double().sum()), float(out["all_trans"].double().sum()),
           (out["all_logits"].clone(), out["all_trans"].clone()) if rank == 0 else None, drv.group_size(), drv.backend_name()))
    drv.barrier()
    drv.close()


@pytest.mark.parametrize("B", [256, 250])
def test_eight_rank_gloo_config4_plan(B):
    """BASELINE config 4's shard plan on CPU: 256 pairs over 8 ranks (32 each - what bench.py --gpus 8 runs per step) and 250
    pairs (32, 32, 31, ... : padded to 32 rows inside the packed buffer, trimmed after the gather).  Every rank runs its own
    shard, enters exactly ONE all-gather, and ends with the single-process result: bit for bit on rank 0, by checksum on the
    others.  (No reference counterpart: train_3DMatch.py:17 pins one GPU)"""
    from gmf_amd.dist import shard_sizes
    world, N = 8, 24
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q, B, N)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _fake_model(_global_batch(B=B, N=N))
    sizes = shard_sizes(B, world)
    assert sizes == ([32] * 8 if B == 256 else [32, 32, 31, 31, 31, 31, 31, 31])
    s_l, s_t = float(ref["logits"].double().sum()), float(ref["final_trans"].double().sum())
    for rank, local, n_coll, t, cl, ct, tensors, gsize, backend in got:
        assert local == sizes[rank] and n_coll == 1 and t == float(world)
        assert cl == s_l and ct == s_t
        assert gsize == world and backend == "gloo"
        if tensors is not None:
            assert torch.equal(tensors[0], ref["logits"]) and torch.equal(tensors[1], ref["final_trans"])


def test_shard_sizes_match_ranges():
    from gmf_amd.dist import shard_range, shard_sizes
    assert shard_sizes(5, 2) == [3, 2] and shard_sizes(256, 8) == [32] * 8 and shard_sizes(3, 4) == [1, 1, 1, 0]
    assert sum(shard_sizes(257, 8)) == 257
