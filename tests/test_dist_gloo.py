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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from gmf_amd.dist import ShardedBatchDriver, shard_batch
    drv = ShardedBatchDriver(_fake_model, world, rank, torch.device("cpu"), backend="gloo")
    # 6 pairs over 2 ranks (all_gather_into_tensor needs equal shards)
    data = shard_batch(_global_batch(B=6), world, rank)
    out = drv.step(data)
    drv.barrier()
    t = drv.max_over_ranks(float(rank + 1))
    q.put((rank, out["all_logits"].clone(), out["all_trans"].clone(), t))
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


def test_two_rank_gloo_matches_single_process():
    world, port = 2, 29731
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_model(_global_batch(B=6))
    for rank, all_logits, all_trans, t in got:
        assert torch.equal(all_logits, ref["logits"])
        assert torch.equal(all_trans, ref["final_trans"])
        assert t == 2.0          # MAX over ranks of (rank + 1)
