"""bench.py - correspondences/second of the GMF multimodal-fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sweep]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both launch lines work for N > 1: under torch.distributed.run every process is one rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
from the environment); without it, `python bench.py --gpus N` starts its N ranks itself - as child processes of a parent that has
not touched the GPU (launch_ranks) - and returns rank 0's JSON line and the worst exit code.  A run with N > 1 whose ranks did
not end up on N DISTINCT devices behind RCCL ("nccl") is not a measurement: it prints the line to stderr with the reason and
exits 3 (--rehearsal, which shares devices over gloo on purpose, says so in the line instead).

One "step" = one pass of the whole hot path (PointDSC.forward in test mode: Fusion-1, 12 x {PointCN,
spatial-consistency attention, Fusion-2 with LCPE}, classifier head, pose head with on-device SVD and
refinement) over one batch of synthetic scene pairs already resident in HBM.  Workload at every N:
BASELINE.json configs[1] per GPU (32 pairs x 5000 correspondences x 128-d, 196 image tokens), fp32 - weak
scaling: pairs are independent, each rank owns its own 32 pairs, and the only exchange is ONE RCCL
all-gather of the packed per-pair logits and poses per step (inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     - the dominant kernel (the spatial-consistency attention) against the f16 MFMA peak / 3 partial products,
                 timed in situ with HIP events on the stream it is launched on; HBM traffic per launch from the committed
                 rocprofv3 PMC summary ONLY when that summary was taken on this very source tree (hash match), else null
  step         - the whole step against the same peak (SURVEY.md section 8d F_logits)
  cpu_baseline - the CPU oracle (a port of the reference's PyTorch CPU path) timed on this host's cores, all cores and 1 thread
  ranks        - per-rank step time and the all-gather's own time (multi-GPU diagnostics)
  sweep        - with --sweep: N = 1000 / 10000, the KITTI shape and the B = 1 latency points (headline unchanged)
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import glob
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it for N > 1)

import torch                                     # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, "Chip-level parameters": dense MFMA peaks, HBM
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
PRODUCTS = 3            # MFMA products per algorithmic multiply-add: split-fp16 operands, hh + hl + lh
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r05_scattn_h2p_pmc.json")
TILE_CYCLES = 1024.0    # matrix-pipe cycles of one key tile per wave in the default form: 16 f16 MFMAs of 32 cycles + 8 block-scaled fp8 MFMAs of 64


def scattn_flops_per_launch(B: int, N: int) -> float:
    """Algorithmic FLOPs of one attention launch: QK^T + PV = 512*N^2 and the fused
    fc_message 128->64->64->128 = 40 960*N, per pair (2 FLOP per MAC; compat/softmax elementwise work excluded)."""
    return B * (512.0 * N * N + 40960.0 * N)


def scattn_bytes_per_launch(B: int, N: int) -> float:
    """Compulsory HBM bytes of one attention launch: Q', K, V, fusion2_out in and the block output (5 x 512 B per row;
    K / V re-reads hit L2 by construction of the XCD-aware grid) - SURVEY.md section 8d.  The compat cache this build streams
    on top of that (4 B per (i, j), padded to 32 x 32 tiles) is NOT algorithmic: it shows up in `traffic`."""
    return B * N * 5 * 512.0


def step_flops(B: int, N: int, T: int) -> float:
    """F_logits(N, T) of SURVEY.md section 8d, per batch."""
    per_pair = 12 * (598784.0 * N + 512.0 * N * N + 256.0 * N * T + 33536.0 * T) + 11840.0 * N + 458752.0 * T + 256.0 * T * T
    return B * per_pair


def csrc_sha16() -> str:
    """Hash of the encoder kernel sources (the attention and linear-stage kernels and their shared headers): ties a committed
    PMC summary to the kernels it was taken on."""
    h = hashlib.sha256()
    for name in ("encoder_kernels.hip", "encoder_h2.hip", "enc_common.hpp", "enc_ff.hpp", "mfma_core.hpp"):
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "gmf_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def build_model(dev, kind):
    import gmf_amd
    from gmf_amd import synthetic
    sigma_d, tau = (0.10, 0.10) if kind == "3dmatch" else (1.2, 1.2)
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=sigma_d)
    model = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                             inlier_threshold=tau, sigma_d=sigma_d, k=40, nms_radius=tau)
    model.load_state_dict(sd, strict=False)
    return model.to(dev).eval(), sd, tau


def make_batch(dev, seeds, N, T, kind):
    from gmf_amd import synthetic
    batch = synthetic.synthetic_batch(seeds, N=N, T=T, kind=kind)
    data = {k: batch[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    return batch, data


def time_steps(driver, data, steps, warmup):
    for _ in range(warmup):
        out = driver.step(data)
    torch.cuda.synchronize()
    driver.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = driver.step(data)
    torch.cuda.synchronize()
    driver.barrier()
    return time.perf_counter() - t0, out


def launch_ranks(n_ranks: int, argv, rehearsal: bool) -> int:
    """`python bench.py --gpus N` without a launcher: one child process per rank, started BEFORE this process makes any GPU call
    (it never makes one: children are new interpreters, nothing is re-executed in a process that initialised HIP).  Rendezvous on
    127.0.0.1 and a free port.  Rank 0 inherits stdout (the ONE JSON line); the parent returns the worst exit code, and when a
    rank dies it ends the ranks that are left by their exact PIDs after a grace period (they would otherwise wait in a
    collective until its timeout)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()            # (counts devices without initialising the runtime on this image)
    if not rehearsal and n_dev < n_ranks:
        print(f"bench.py --gpus {n_ranks}: this host shows {n_dev} HIP device(s); one rank per GPU needs {n_ranks} "
              "(--rehearsal shares devices over gloo and says so in its line)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, deadline = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        rcs = [p.poll() for p in procs]
        if deadline is None and any(rc not in (None, 0) for rc in rcs):
            deadline = time.time() + 30.0        # a rank failed: the others get 30 s to leave their collective and report
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                     # (this process's own children, by PID)
    for p in procs:
        rc = p.wait()
        worst = worst or (rc if rc > 0 else (128 - rc if rc < 0 else 0))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=32, help="scene pairs per GPU per step")
    ap.add_argument("--ncorr", type=int, default=5000)
    ap.add_argument("--tokens", type=int, default=196)
    ap.add_argument("--kind", default="3dmatch", choices=["3dmatch", "kitti"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="full sweep: the default size points (N = 1000 / 10000, KITTI shape, B = 1, ragged) plus the "
                                                       "throughput numerics modes and the DGR rows (rank 0, N = 1 GPU)")
    ap.add_argument("--no-sweep", action="store_true", help="headline only")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 ranks on a box with fewer GPUs: ranks share the devices (local_rank %% device_count) and exchange "
                         "through gloo staged over the host.  Exercises the launch line, the sharded step and the JSON line; the "
                         "numbers are NOT a measurement (the line says so)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.rehearsal))     # no launcher: start the ranks ourselves
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the HIP path is mandatory, there is no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count() if args.rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from gmf_amd import _lib
    from gmf_amd.dist import ShardedBatchDriver

    B, N, T = args.pairs, args.ncorr, args.tokens
    model, sd, tau = build_model(dev, args.kind)
    # (communication libraries announce themselves on stdout when a group forms - gloo: "[Gloo] Rank 0 is connected to ..." - and stdout
    # carries exactly ONE JSON line: file descriptor 1 points at stderr while the group is set up)
    sys.stdout.flush()
    fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        driver = ShardedBatchDriver(model, world, rank, dev, backend="gloo" if args.rehearsal else None)
        driver.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(fd1, 1)
        os.close(fd1)
    seeds = [rank * B + i for i in range(B)]                 # every rank owns its own pairs (weak scaling)
    batch, data = make_batch(dev, seeds, N, T, args.kind)
    torch.cuda.synchronize()

    driver.check_status = False          # (the status column of the packed rows is read once, after the timed region)
    for _ in range(args.warmup):
        out = driver.step(data)
    torch.cuda.synchronize()

    handle = _lib.handle_for(dev_index)
    handle.call("gmf_profile_enable", 1)
    driver.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = driver.step(data)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    driver.barrier()
    dt = time.perf_counter() - t0
    ms_total, launches = C.c_double(0.0), C.c_int(0)
    handle.call("gmf_profile_read", C.byref(ms_total), C.byref(launches))
    handle.call("gmf_profile_enable", 0)
    dt = driver.max_over_ranks(dt)
    # diagnostics outside the timed region: per-rank step time, and one more step with the model / gather split timed by events
    per_rank_ms = driver.gather_floats(t_local / args.steps * 1e3)
    driver.time_steps = True
    driver.check_status = True           # any rank that failed during the run raises here, on every rank
    driver.step(data)
    model_ms, gather_ms = driver.read_timings()
    driver.time_steps = False
    gather_all = driver.gather_floats(gather_ms if gather_ms is not None else 0.0)
    devices = driver.gather_objects(device_identity(dev_index))

    if rank != 0:
        driver.close()
        return

    value = world * B * N * args.steps / dt
    avg_ms = ms_total.value / max(1, launches.value)
    flops = scattn_flops_per_launch(B, N)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    peak = PEAK_F16_MFMA_TFLOPS / PRODUCTS
    alg_bytes = scattn_bytes_per_launch(B, N)
    traffic, traffic_note = None, "no PMC summary for this source tree (profiles/r05_scattn_h2p_pmc.json missing or taken on another build)"
    sha = csrc_sha16()
    if os.path.exists(PMC_SUMMARY):
        pmc = json.load(open(PMC_SUMMARY))
        wl = pmc.get("workload", {})
        if pmc.get("csrc_sha16") == sha and (wl.get("pairs"), wl.get("n_corr"), wl.get("tokens")) == (B, N, T):
            traffic = pmc["derived"]["traffic_bytes_per_launch"]
            traffic_note = f"rocprofv3 FETCH_SIZE / WRITE_SIZE passes on this source tree (csrc_sha16 {sha}), profiles/r05_scattn_h2p_pmc.json"
    step_ms = dt / args.steps * 1e3
    step_tflops = step_flops(B, N, T) / (step_ms * 1e-3) / 1e12
    line = {
        "metric": "correspondences/sec (whole node)", "value": value, "unit": "correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (split-fp16 MFMA operands, fp32 accumulate; the cross products of the attention's two contractions, S = K Q'^T and O = P V, on block-scaled e4m3 operands wherever the device-side score guard allows)", "data": "synthetic",
        "config": {"workload": f"synthetic {args.kind}-shape pairs, PointDSC.forward test mode (logits + R,t)",
                   "pairs_per_gpu": B, "global_pairs": world * B, "n_corr": N, "feat_dim": 128, "image_tokens": T,
                   "layers": 12, "parallelism": f"pairs sharded over {world} GPU(s), one RCCL all-gather of packed logits+poses"},
        "roofline": {"bound": "mfma", "kernel": "k_scattn_h2p (spatial-consistency attention + fc_message)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_note": traffic_note,
                     "algorithmic_bytes": alg_bytes,
                     "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                     "hbm_floor_ms": (traffic / (PEAK_HBM_GBS * 1e9) * 1e3) if traffic else None,
                     "hbm_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                     "hbm_frac_of_peak": (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None,
                     "peak_note": (f"f16 MFMA dense peak {PEAK_F16_MFMA_TFLOPS:.0f} TFLOP/s / {PRODUCTS} partial products per algorithmic "
                                   "multiply-add (split-fp16 operands, fp32 accumulate, fp32-equivalent results) - the yardstick of rounds "
                                   "1-2, kept.  Since round 3 the two cross products of O += P V, since round 5 those of S = K Q'^T as "
                                   "well, run on the block-scaled fp8 pipe wherever the device-side score guard allows: a tile takes "
                                   "16 f16 + 8 fp8 matrix instructions = 1024 matrix-pipe cycles instead of 48 x 32 = 1536"),
                     # matrix-pipe cycles the kernel needs per algorithmic flop, as a fraction of the cycles it had ("pipe_busy" if
                     # nothing else stalled it): 1024 of 1536 cycles per tile
                     "frac_of_pipe_cycles": achieved / (peak * 1536.0 / TILE_CYCLES),
                     "executed_mfma_tflops": achieved * PRODUCTS * TILE_CYCLES / 1536.0,
                     "executed_note": "f16-equivalent matrix-pipe work per second (an fp8 MFMA of 64 cycles counted as two f16 MFMAs of 32)",
                     "avg_launch_ms": avg_ms, "launches_timed": launches.value, "flops_per_launch": flops},
        "step": {"algorithmic_tflops": step_tflops, "frac_of_peak": step_tflops / peak,
                 "flops_per_step": step_flops(B, N, T), "note": "F_logits of SURVEY.md section 8d x pairs / step time"},
        "ranks": {"per_rank_ms_per_step": per_rank_ms, "model_ms": model_ms, "all_gather_ms": gather_ms,
                  "all_gather_ms_per_rank": gather_all,
                  "devices": devices, "distinct_devices": len({d["pci"] + "/" + d["uuid"] for d in devices}),
                  "nranks": driver.group_size(), "backend": driver.backend_name(),
                  "note": "outside the timed region: one extra step with HIP events around the forward and around pack + all-gather"},
        "csrc_sha16": sha,
    }
    if args.rehearsal:
        line["rehearsal"] = "ranks share the GPU(s) and exchange through gloo over the host: NOT a measurement"
    elif world > 1:
        rk = line["ranks"]
        why = []
        if rk["distinct_devices"] != world or rk["nranks"] != world:
            why.append(f"{world} ranks on {rk['distinct_devices']} distinct device(s), process group of {rk['nranks']}")
        if rk["backend"] != "nccl":
            why.append(f"backend {rk['backend']} instead of nccl (RCCL)")
        if why:
            line["invalid"] = "; ".join(why) + ": not a multi-GPU measurement"
            print(json.dumps(line), file=sys.stderr)
            driver.close()
            sys.exit(3)
    if world == 1 and not args.no_sweep:
        # the size points of SURVEY.md section 8d ride in the default line too (~10 s), so that they are in the DRIVER's record
        # and not only in builder-run files; --sweep adds the throughput modes and the DGR rows
        line["sweep"] = sweep(dev, full=args.sweep)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"], line["parity"] = cpu_baseline(sd, batch, out, N, T, tau)
        if args.kind == "3dmatch":
            line["parity"]["kitti_shape"] = kitti_parity(dev)
    print(json.dumps(line))
    driver.close()


def sweep(dev, full=False):
    """Other operating points of the same build: N in {1 k, 10 k}, the KITTI shape (config 3), the
    B = 1 latency points (the reference's evaluation mode) and one ragged batch.  Few steps each; the headline is not affected.
    full: also the throughput numerics modes and the DGR rows."""
    from gmf_amd.dist import ShardedBatchDriver
    rows = []
    peak = PEAK_F16_MFMA_TFLOPS / PRODUCTS
    # (the T = 300 row: the reference's real token count - 120 x 160 images through ResNet-34's layer2 give 15 x 20 tokens,
    # config_3DMatch.py:105-106, resnet.py:198-216; BASELINE configs[1] names 196)
    for kind, B, N, T, steps in (("3dmatch", 32, 1000, 196, 5), ("3dmatch", 32, 10000, 196, 2), ("kitti", 16, 10000, 196, 2),
                                 ("3dmatch", 32, 5000, 300, 3),
                                 ("3dmatch", 1, 1000, 196, 20), ("3dmatch", 1, 5000, 196, 10), ("3dmatch", 1, 10000, 196, 5)):
        model, _, _ = build_model(dev, kind)
        _, data = make_batch(dev, list(range(B)), N, T, kind)
        drv = ShardedBatchDriver(model, 1, 0, dev)
        gc.collect()
        torch.cuda.synchronize()
        # best of three timed batches of steps: the loop also frees the previous configuration's model (Python GC, hipFree of
        # its blobs), which can land as one ~70 ms stall anywhere in a batch of sub-millisecond steps
        dt = min(time_steps(drv, data, steps, 2)[0] for _ in range(3))
        ms = dt / steps * 1e3
        tf = step_flops(B, N, T) / (ms * 1e-3) / 1e12
        rows.append({"workload": f"{kind} {B} pairs x {N}" + ("" if T == 196 else f", {T} image tokens"), "ms_per_step": ms, "value": B * N / (ms * 1e-3),
                     "unit": "correspondences/s", "step_frac_of_peak": tf / peak})
        del model, data
        torch.cuda.empty_cache()
    # ragged batch: 32 pairs whose N is drawn from U[4000, 5500] (what the reference's evaluation loop sees pair by pair,
    # evaluation/test_3DMatch.py:69) in ONE launch (gmf_encoder_forward_ragged + gmf_pose_head_ragged), against the same pairs
    # fed one B = 1 call at a time
    import numpy as np
    from gmf_amd import synthetic
    model, _, _ = build_model(dev, "3dmatch")
    sizes = [int(n) for n in np.random.default_rng(77).integers(4000, 5501, size=32)]
    pairs = [synthetic.synthetic_batch([900 + i], N=n, T=196) for i, n in enumerate(sizes)]
    rag = {k: [b[k][0].to(dev) for b in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag["p_tokens"] = torch.cat([b["p_tokens"] for b in pairs]).to(dev)
    rag["q_tokens"] = torch.cat([b["q_tokens"] for b in pairs]).to(dev)
    packed = {k: torch.cat(rag[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    packed.update(p_tokens=rag["p_tokens"], q_tokens=rag["q_tokens"], n_points=sizes, testing=True)
    singles = [{**{k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}, "testing": True} for b in pairs]

    def timed(fn, n):
        for _ in range(2):
            fn()
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best

    ms_r = timed(lambda: model(packed), 5)
    ms_1 = timed(lambda: [model(d) for d in singles], 2)
    rows.append({"workload": "3dmatch 32 pairs, N ~ U[4000, 5500] (sum %d), ONE ragged launch" % sum(sizes), "ms_per_step": ms_r,
                 "value": sum(sizes) / (ms_r * 1e-3), "unit": "correspondences/s",
                 "same_pairs_as_32_calls_with_B_1": {"ms": ms_1, "value": sum(sizes) / (ms_1 * 1e-3)}})
    # ... and a SMALL ragged batch: four requests with their own N, as a serving process would batch them (the three-launch small-grid
    # form with the pair table, round 5)
    sizes4 = [int(n) for n in np.random.default_rng(78).integers(800, 1201, size=4)]
    packed4 = {k: torch.cat([b[k][0][:n].to(dev) for b, n in zip(pairs[:4], sizes4)]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    packed4.update(p_tokens=rag["p_tokens"][:4], q_tokens=rag["q_tokens"][:4], n_points=sizes4, testing=True)
    singles4 = [{**{k: (d[k][:, :n] if k in ("corr_pos", "src_keypts", "tgt_keypts") else d[k]) for k in d if k != "testing"}, "testing": True}
                for d, n in zip(singles[:4], sizes4)]
    ms_r4 = timed(lambda: model(packed4), 20)
    ms_14 = timed(lambda: [model(d) for d in singles4], 5)
    rows.append({"workload": "3dmatch 4 pairs, N ~ U[800, 1200] (sum %d), ONE ragged launch" % sum(sizes4), "ms_per_step": ms_r4,
                 "value": sum(sizes4) / (ms_r4 * 1e-3), "unit": "correspondences/s",
                 "same_pairs_as_4_calls_with_B_1": {"ms": ms_14, "value": sum(sizes4) / (ms_14 * 1e-3)}})
    del model, rag, packed, singles, pairs, packed4, singles4
    torch.cuda.empty_cache()
    rows += dgr_rows(dev, full)
    # the throughput numerics modes (gmf_set_tuning "precision" = 1, 2; NOT the parity path, never the headline) on the headline
    # workload, with its measured deviation from the parity mode on the same batch
    from gmf_amd import _lib
    model, _, _ = build_model(dev, "3dmatch")
    _, data = make_batch(dev, list(range(32)), 5000, 196, "3dmatch")
    drv = ShardedBatchDriver(model, 1, 0, dev)
    hnd = _lib.handle_for(dev.index or 0)
    res0 = model(data)
    lg0, T0 = model.last_logits.clone(), res0["final_trans"].clone()
    labels = {1: "throughput numerics (precision = 1: fp16 one-product attention, fp16 compat)",
              2: "throughput numerics, level 2 (precision = 2: level 1 + one-product linear stages)"}
    # (level 1 rides in the DEFAULT line - BASELINE configs 2-3 name bf16-class arithmetic; it is labelled outside the parity gate - level 2
    # only with --sweep)
    for level in ((1, 2) if full else (1,)):
        try:
            hnd.call("gmf_set_tuning", b"precision", level)
            res1 = model(data)
            dl, dT = float((model.last_logits - lg0).abs().max()), float((res1["final_trans"] - T0).abs().max())
            dt = min(time_steps(drv, data, 5, 2)[0] for _ in range(3))
        finally:
            hnd.call("gmf_set_tuning", b"precision", 0)
        ms = dt / 5 * 1e3
        rows.append({"workload": "3dmatch 32 pairs x 5000, " + labels[level],
                     "ms_per_step": ms, "value": 32 * 5000 / (ms * 1e-3), "unit": "correspondences/s",
                     "max_abs_dlogit_vs_parity_mode": dl, "max_abs_dT_vs_parity_mode": dT, "within_parity_gate": False})
    del model, data
    torch.cuda.empty_cache()
    return rows


def dgr_rows(dev, full):
    """The DGR plugin surface in the driver's own line: BASELINE config 5 (batched weighted-SVD pose, 32 problems x 8000
    correspondences, core/registration.py:91-113) with its HBM fraction - it is latency-bound and the row says so - and the
    bottleneck PerceiverIO of row a15 at 20 000 voxels (model/resunet_new.py:516-525).  full: also GlobalRegistration and the
    smaller voxel counts."""
    rows = []
    peak = PEAK_F16_MFMA_TFLOPS / PRODUCTS
    # the DGR plugin surface (BASELINE config 5: batched weighted-SVD pose, N = 8000; and the bottleneck PerceiverIO of row a15)
    import numpy as np
    import gmf_amd
    from gmf_amd import synthetic

    def best_ms(fn, n):
        for _ in range(2):
            fn()
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best

    Bd, Nd = 32, 8000
    scenes = [synthetic.dgr_scene(Nd, 500 + i) for i in range(Bd)]
    X = torch.cat([sc[0] for sc in scenes]).to(dev)
    Y = torch.cat([sc[1] for sc in scenes]).to(dev)
    wts = torch.cat([sc[2] for sc in scenes]).to(dev)
    off = [i * Nd for i in range(Bd + 1)]
    ms = best_ms(lambda: gmf_amd.weighted_procrustes_batched(X, Y, wts, off, np.finfo(np.float32).eps), 20)
    wp_bytes = Bd * (28.0 * Nd + 48.0)               # 28 B per correspondence in, 48 B out per pair
    rows.append({"workload": "dgr weighted_procrustes, 32 problems x 8000 correspondences (config 5)", "ms_per_step": ms,
                 "value": Bd * Nd / (ms * 1e-3), "unit": "correspondences/s",
                 "roofline": {"bound": "hbm", "achieved": wp_bytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": wp_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                              "note": "latency-bound: 7 MB per launch, one pass of one workgroup per pair and an fp64 3x3 SVD on one lane, behind ~15 us of host-side call overhead; the HBM figure is what the size implies, not what limits it"}})
    if full:
        ms = best_ms(lambda: gmf_amd.global_registration_batched(X, Y, wts, off, break_threshold_ratio=1e-4, quantization_size=0.1), 5)
        rows.append({"workload": "dgr GlobalRegistration (Adam refinement to convergence), 32 problems x 8000 correspondences", "ms_per_step": ms,
                     "value": Bd * Nd / (ms * 1e-3), "unit": "correspondences/s"})
    # row f-2: descriptor nearest-neighbour matching, the step that produces the putative correspondences (ThreeDMatch.py:164-166):
    # fused distance GEMM on the f32 MFMA (exact f32 products) + row argmin, nothing of size N x N written
    Nm, dm = 5000, 32
    fa = torch.nn.functional.normalize(torch.randn(Nm, dm, device=dev), dim=1)
    fb = torch.nn.functional.normalize(torch.randn(Nm, dm, device=dev), dim=1)
    ms = best_ms(lambda: gmf_amd.nn_match(fa, fb), 20)
    tf = 2.0 * dm * Nm * Nm / (ms * 1e-3) / 1e12
    rows.append({"workload": f"descriptor matching (row f-2), {Nm} x {Nm} descriptors x {dm}-d, whole call (one preparation launch - both images, norms, identity of the minimum -, match, winners)",
                 "ms_per_step": ms, "value": Nm / (ms * 1e-3), "unit": "source descriptors/s", "algorithmic_tflops": tf,
                 "roofline": {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                              "note": "f32 MFMA dense peak (v_mfma_f32_32x32x2_f32); the match kernel alone is 22 us of the call"}})
    pio = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                              latent_dim_head=64, pe=True).to(dev).eval()
    for M in ((1000, 4000, 20000) if full else (20000,)):
        xq, img = torch.randn(1, M, 256, device=dev), torch.randn(1, 300, 128, device=dev)
        ms = best_ms(lambda: pio(img, queries_encoder=xq), 10)
        tf = M * (1713152 + 512 * 300) / (ms * 1e-3) / 1e12
        rows.append({"workload": f"dgr bottleneck PerceiverIO (256 wide, head 128), {M} voxels x 300 image tokens", "ms_per_step": ms,
                     "value": M / (ms * 1e-3), "unit": "voxels/s", "algorithmic_tflops": tf,
                     "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}})
    return rows


def device_identity(index: int) -> dict:
    """What tells one GPU of the node from another: name, PCI address and UUID of this rank's device, its host process."""
    p = torch.cuda.get_device_properties(index)
    pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return {"rank": int(os.environ.get("RANK", "0")), "device_index": index, "name": p.name, "pci": pci,
            "uuid": str(getattr(p, "uuid", "")), "cus": p.multi_processor_count, "pid": os.getpid()}


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


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, batch, gpu_out, N, T, tau):
    """Time the CPU oracle (port of the reference's PyTorch CPU path) on a bounded sample: pair 0 of the same
    batch, test mode, B=1 as the reference runs it; 1 warm-up + 5 repetitions on all cores (~12 s of CPU work), median,
    then 1 repetition on ONE thread (~15 s; scaling context).  Also report parity."""
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
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        O.pointdsc_forward(sd, one, inlier_threshold=tau, nms_radius=tau, testing=True)
        t1 = time.perf_counter() - t0
        torch.set_num_threads(cores)
    med = statistics.median(ts)
    base = {"value": N / med, "unit": "correspondences/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "value_1_thread": N / t1,
            "sample": f"1 pair x {N} correspondences x 128-d, {T} tokens, PointDSC.forward test mode, median of 5 "
                      f"after 1 warm-up on {cores} threads ({med:.2f} s per pair) + 1 run on 1 thread ({t1:.1f} s), "
                      f"torch {torch.__version__} CPU fp32"}
    parity = {"max_abs_dlogit": float((gpu_out["logits"][:1].cpu() - ref["logits"]).abs().max()),
              "max_abs_dT": float((gpu_out["final_trans"][:1].cpu() - ref["final_trans"]).abs().max()),
              "vs": "CPU oracle on pair 0 of the timed batch"}
    # which of the two fp32 evaluations is closer to the exact network: an fp64 evaluation of the same encoder on the same pair
    truth = fp64_logits(sd, one, float(sd["sigma_d"]) if "sigma_d" in sd else 0.10)
    parity["max_abs_dlogit_vs_fp64"] = {"hip": float((gpu_out["logits"][:1].cpu().double() - truth).abs().max()),
                                        "fp32_oracle": float((ref["logits"].double() - truth).abs().max()),
                                        "note": "fp64 evaluation of the same encoder + classifier (oracle code, float64) on pair 0"}
    return base, parity


def fp64_logits(sd, one, sigma_d):
    """Inlier logits of the oracle's encoder + classifier evaluated in float64 (the reference network without fp32 rounding)."""
    from oracle import gmf_oracle as O
    sd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in one.items()}
    with torch.no_grad():
        compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], sigma_d)
        return O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))


def kitti_parity(dev):
    """The same pair of numbers for ONE KITTI-shape pair (config 3: N = 10000, sigma_d = 1.2, +-40 m coordinates): HIP and the
    fp32 oracle against the fp64 evaluation, and against each other - for the seeded weights as they are (scaled for
    3DMatch-size coordinates: the reference's own fp32 is 2e-3 from the exact network there, a stress case) and, under
    `conditioned`, for `synthetic.kitti_conditioned` (layer0.weight / 13: activations at the 3DMatch scale), the weight set on
    which the literal 1e-4 gate is meaningful (golden F22 pins the oracle on this branch against the reference itself)."""
    from oracle import gmf_oracle as O
    from gmf_amd import synthetic
    out = None
    for name in ("stress", "conditioned"):
        model, sd, tau = build_model(dev, "kitti")
        if name == "conditioned":
            sd = synthetic.kitti_conditioned(sd)
            model.load_state_dict(sd, strict=False)
            model = model.to(dev).eval()
        batch, data = make_batch(dev, [0], 10000, 196, "kitti")
        from gmf_amd import _lib
        _lib.handle_for(dev.index or 0).status(clear=True)
        res = model(data)
        torch.cuda.synchronize()
        lg = model.last_logits.cpu()
        with torch.no_grad():
            ref = O.pointdsc_forward(sd, batch, inlier_threshold=tau, nms_radius=tau, testing=True)
        truth = fp64_logits(sd, batch, 1.2)
        e_hip, e_ref = float((lg.double() - truth).abs().max()), float((ref["logits"].double() - truth).abs().max())
        rec = {"max_abs_dlogit": float((lg - ref["logits"]).abs().max()),
               "max_abs_dT": float((res["final_trans"].cpu() - ref["final_trans"]).abs().max()),
               "max_abs_dlogit_vs_fp64": {"hip": e_hip, "fp32_oracle": e_ref},
               # the ONE floor-relative contract of the default numerics on every weight set (tests: test_f22_kitti_branch):
               # no further from the fp64 evaluation than 1.5 x the reference's own fp32 evaluation + 2e-5
               "floor_relative": {"bound": 1.5 * e_ref + 2e-5, "hip_over_fp32_oracle": e_hip / max(e_ref, 1e-30),
                                  "within": bool(e_hip < 1.5 * e_ref + 2e-5)},
               "pv_fp8_guard_tripped": bool(_lib.handle_for(dev.index or 0).status() & _lib.GMF_STATUS_PV_GUARDED)}
        if out is None:
            out = {"workload": "1 kitti-shape pair x 10000 correspondences (config 3), sigma_d 1.2", **rec}
        else:
            rec["weights"] = "synthetic.kitti_conditioned: encoder.layer0.weight / 13 (golden F22)"
            rec["gate"] = 1e-4
            rec["within_gate"] = bool(rec["max_abs_dlogit"] < 1e-4)
            out[name] = rec
    return out


if __name__ == "__main__":
    main()
