"""Where the time of a latency-bound stage-chain kernel goes, without instrumenting it: k_fusion_attn_w_h2 built with an early
exit after each phase (a patched COPY of fusion_wide.hip, -DTRUNC=k; the exit stores a value that depends on everything
computed so far, so nothing is optimised away).  Kernel time per variant under the torch profiler = cumulative timeline.
(Stamps inside the kernel - s_memrealtime after every stage, tools/ubench/stage_timeline.py - doubled its run time.)

    python tools/ubench/stage_truncate.py build      # here
    python tools/ubench/stage_truncate.py run [M]    # GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
PHASES = ["loads + barrier", "LCPE", "LayerNorm + split", "to_q (8 stages)", "10 context tiles (20 stages)", "normalise", "whole kernel"]


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    work = os.path.join(OUT, "src_trunc")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip")):
            shutil.copy(os.path.join(CSRC, f), work)
    path = os.path.join(work, "fusion_wide.hip")
    text = open(path).read()
    a = text.index("k_fusion_attn_w_h2(const float*")
    a = text.rindex("template <bool PE>", 0, a)
    b = text.index("\n}\n", a) + 3
    k = text[a:b]

    def rep(old, new):
        nonlocal k
        assert k.count(old) == 1, (old, k.count(old))
        k = k.replace(old, new)
    sink = "{ float sk = 0.f; for (int e = 0; e < LATF; ++e) sk += xp[e]; %s asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); if (active) x1_out[toff + lane] = sk; return; }"
    rep("  if (PE) LcpeHalo<LATF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);\n",
        "  if (TRUNC == 0) " + sink % "" + "\n  if (PE) LcpeHalo<LATF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);\n  if (TRUNC == 1) " + sink % "" + "\n")
    rep("      nx.set(xn);\n    }\n",
        "      nx.set(xn);\n    }\n    if (TRUNC == 2) " + sink % "for (int s = 0; s < 16; ++s) sk += (float)nx.h[s][0] + (float)nx.l[s][1];" + "\n")
    rep("  f32x16 oacc[4];\n", "  if (TRUNC == 3) " + sink % "for (int s = 0; s < 8; ++s) sk += (float)qx.h[s][0] + (float)qx.l[s][1];" + "\n  f32x16 oacc[4];\n")
    rep("  FragH2<8> ox;\n", "  if (TRUNC == 4) " + sink % "for (int db = 0; db < 4; ++db) for (int r = 0; r < 16; ++r) sk += oacc[db][r]; sk += l_half;" + "\n  FragH2<8> ox;\n")
    rep("#pragma unroll\n  for (int mb = 0; mb < 8; ++mb) {\n    const f16x8* lw = as_h2(ss.acquire());\n    f32x16 acc = zero16();\n    mma_wx_h2<8>(acc, lw, ox);",
        "  if (TRUNC == 5) " + sink % "for (int s = 0; s < 8; ++s) sk += (float)ox.h[s][0] + (float)ox.l[s][1];" +
        "\n#pragma unroll\n  for (int mb = 0; mb < 8; ++mb) {\n    const f16x8* lw = as_h2(ss.acquire());\n    f32x16 acc = zero16();\n    mma_wx_h2<8>(acc, lw, ox);")
    text = text[:a] + k + text[b:]
    open(path, "w").write(text)
    rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o != "fusion_wide.o"]
    for t in range(len(PHASES)):
        obj = os.path.join(work, f"fusion_wide_{t}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-fno-slp-vectorize", f"-DTRUNC={t}", "-c", path, "-o", obj])
        lib = os.path.join(OUT, f"libgmf_hip_trunc{t}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + rest + ["-o", lib])
        print("built", lib, flush=True)
    shutil.rmtree(work)


def run_one(t, M):
    import torch
    sys.path.insert(0, ROOT)
    from gmf_amd import _lib
    _lib.LIB_PATH = os.path.join(OUT, f"libgmf_hip_trunc{t}.so")
    import gmf_amd
    from torch.profiler import profile, ProfilerActivity
    dev = torch.device("cuda:0")
    pio = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                              latent_dim_head=64, pe=True).to(dev).eval()
    xq = torch.randn(1, M, 256, device=dev)
    img = torch.randn(1, 300, 128, device=dev)
    for _ in range(5):
        pio(img, queries_encoder=xq)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(20):
            pio(img, queries_encoder=xq)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "k_fusion_attn_w_h2" in e.key:
            print(f"  exit after {PHASES[t]:32s}: {e.device_time_total / e.count:6.1f} us", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one(int(sys.argv[2]), int(sys.argv[3]))
    else:
        M = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
        print(f"k_fusion_attn_w_h2<PE>, M = {M}: kernel time when it returns after ...")
        for t in range(len(PHASES)):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(t), str(M)], stderr=subprocess.DEVNULL)
