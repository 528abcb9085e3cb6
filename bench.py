"""bench.py - correspondences/second of the GMF multimodal-fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (PointDSC.forward in test mode: Fusion-1, 12 x {PointCN,
spatial-consistency attention, Fusion-2 with LCPE}, classifier head, pose head with on-device SVD and
refinement) over one batch of synthetic scene pairs already resident in HBM.  Workload at every N:
BASELINE.json configs[1] per GPU (32 pairs x 5000 correspondences x 128-d, 196 image tokens), fp32 - weak
scaling: pairs are independent, each rank owns its own 32 pairs, and the only exchange is one RCCL
all-gather of the per-pair logits and poses per step (inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (the spatial-consistency attention) against the f16 MFMA peak / 3 partial products,
                 timed in situ with HIP events on the stream it is launched on
  cpu_baseline - the CPU oracle (a port of the reference's PyTorch CPU path) timed on this host's cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, "Chip-level parameters": dense MFMA peaks
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
# MFMA products issued per algorithmic multiply-add by each form of the attention kernel (gmf_set_tuning "scattn_variant")
PRODUCTS = {0: 1, 1: 1, 2: 1, 3: 6, 4: 6, 5: 6, 6: 6, 7: 6, 8: 6, 9: 3, 10: 3, 16: 3, 17: 3, 18: 3}


def scattn_flops_per_launch(B: int, N: int) -> float:
    """Algorithmic FLOPs of one k_scattn launch: QK^T + PV = 512*N^2 and the fused
    fc_message 128->64->64->128 = 40 960*N, per pair (2 FLOP per MAC; compat/softmax elementwise work excluded)."""
    return B * (512.0 * N * N + 40960.0 * N)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=32, help="scene pairs per GPU per step")
    ap.add_argument("--ncorr", type=int, default=5000)
    ap.add_argument("--tokens", type=int, default=196)
    ap.add_argument("--kind", default="3dmatch", choices=["3dmatch", "kitti"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the HIP path is mandatory, there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import gmf_amd
    from gmf_amd import _lib, synthetic
    from gmf_amd.dist import ShardedBatchDriver

    B, N, T = args.pairs, args.ncorr, args.tokens
    sigma_d, tau = (0.10, 0.10) if args.kind == "3dmatch" else (1.2, 1.2)
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=sigma_d)
    model = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                             inlier_threshold=tau, sigma_d=sigma_d, k=40, nms_radius=tau)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()

    driver = ShardedBatchDriver(model, world, rank, dev)
    seeds = [rank * B + i for i in range(B)]                 # every rank owns its own pairs (weak scaling)
    batch = synthetic.synthetic_batch(seeds, N=N, T=T, kind=args.kind)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = driver.step(data)
    torch.cuda.synchronize()

    handle = _lib.handle_for(local_rank)
    handle.call("gmf_profile_enable", 1)
    driver.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = driver.step(data)
    torch.cuda.synchronize()
    driver.barrier()
    dt = time.perf_counter() - t0
    import ctypes as C
    ms_total, launches = C.c_double(0.0), C.c_int(0)
    handle.call("gmf_profile_read", C.byref(ms_total), C.byref(launches))
    handle.call("gmf_profile_enable", 0)
    dt = driver.max_over_ranks(dt)

    if rank != 0:
        driver.close()
        return

    value = world * B * N * args.steps / dt
    avg_ms = ms_total.value / max(1, launches.value)
    achieved = scattn_flops_per_launch(B, N) / (avg_ms * 1e-3) / 1e12
    variant = int(os.environ.get("GMF_SCATTN", "18"))
    nprod = PRODUCTS.get(variant, 3)
    if nprod == 1:
        peak, peak_note, dtype = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense peak", "f32"
    else:
        peak = PEAK_F16_MFMA_TFLOPS / nprod
        peak_note = (f"f16/bf16 MFMA dense peak 2500 TFLOP/s / {nprod} partial products per algorithmic multiply-add "
                     "(split-precision operands, fp32 accumulate, fp32-equivalent results)")
        dtype = "f32 (split-fp16 MFMA operands, fp32 accumulate)" if nprod == 3 else "f32 (split-bf16 MFMA operands, fp32 accumulate)"
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r01_scattn_h2p_pmc.json")
    if variant == 18 and (B, N, T) == (32, 5000, 196) and os.path.exists(pmc_file):
        traffic = json.load(open(pmc_file))["derived"]["traffic_bytes_per_launch"]   # rocprofv3 PMC, see the file
    line = {
        "metric": "correspondences/sec (whole node)", "value": value, "unit": "correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"synthetic {args.kind}-shape pairs, PointDSC.forward test mode (logits + R,t)",
                   "pairs_per_gpu": B, "global_pairs": world * B, "n_corr": N, "feat_dim": 128, "image_tokens": T,
                   "layers": 12, "parallelism": f"pairs sharded over {world} GPU(s), RCCL all-gather of logits+poses"},
        "roofline": {"bound": "mfma", "kernel": "k_scattn (spatial-consistency attention + fc_message)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "peak_note": peak_note,
                     "executed_mfma_tflops": achieved * nprod, "x_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "avg_launch_ms": avg_ms, "launches_timed": launches.value,
                     "flops_per_launch": scattn_flops_per_launch(B, N)},
    }

    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"], line["parity"] = cpu_baseline(sd, batch, out, N, T, tau)
    print(json.dumps(line))
    driver.close()


def host_cpu_share() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, batch, gpu_out, N, T, tau):
    """Time the CPU oracle (port of the reference's PyTorch CPU path) on a bounded sample: pair 0 of the same
    batch, test mode, B=1 as the reference runs it; 1 warm-up + 5 repetitions (~12 s of CPU work), median.  Also report parity."""
    from oracle import gmf_oracle as O
    import statistics
    one = {k: v[:1] for k, v in batch.items()}
    cores = host_cpu_share()
    torch.set_num_threads(cores)
    with torch.no_grad():
        ref = O.pointdsc_forward(sd, one, inlier_threshold=tau, nms_radius=tau, testing=True)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ref = O.pointdsc_forward(sd, one, inlier_threshold=tau, nms_radius=tau, testing=True)
            ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    base = {"value": N / med, "unit": "correspondences/s", "cores": cores, "kind": "port",
            "sample": f"1 pair x {N} correspondences x 128-d, {T} tokens, PointDSC.forward test mode, median of 5 "
                      f"after 1 warm-up ({med:.2f} s per pair), torch {torch.__version__} CPU fp32"}
    parity = {"max_abs_dlogit": float((gpu_out["logits"][:1].cpu() - ref["logits"]).abs().max()),
              "max_abs_dT": float((gpu_out["final_trans"][:1].cpu() - ref["final_trans"]).abs().max()),
              "vs": "CPU oracle on pair 0 of the timed batch"}
    return base, parity


if __name__ == "__main__":
    main()
