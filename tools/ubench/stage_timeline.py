"""Where the time of a latency-bound stage-chain kernel goes: a timeline of k_fusion_attn_w_h2 (one wave per SIMD, 36 weight /
context stages of 16 KiB) from s_memrealtime stamps (100 MHz) taken by wave 0 of workgroup 0 after every stage.  The stamps
live in a patched COPY of fusion_wide.hip built into tools/_ab/libgmf_hip_timeline.so (never in the library).

    python tools/ubench/stage_timeline.py build      # here (cross-compiles)
    python tools/ubench/stage_timeline.py run [M]    # GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
LIB = os.path.join(OUT, "libgmf_hip_timeline.so")

LABELS = (["start", "primed", "lcpe", "ln+split"] + [f"to_q {i}" for i in range(4)] + [f"ctx tile {i}" for i in range(10)] +
          ["normalise"] + [f"to_out {i}" for i in range(8)])


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    work = os.path.join(OUT, "src_timeline")
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

    def rep(old, new, count=1):
        nonlocal k
        assert k.count(old) == count, (old, k.count(old))
        k = k.replace(old, new)
    rep("  StageRing<kRingW> ss;\n", "  int tl_n = 0;\n  TL();\n  StageRing<kRingW> ss;\n")
    rep("  ss.prime();\n", "  ss.prime();\n  TL();\n")
    rep("  FragH2<8> qx;\n", "  TL();\n  FragH2<8> qx;\n")
    rep("      nx.set(xn);\n    }\n", "      nx.set(xn);\n    }\n    TL();\n")
    rep("      qx.set_block(mb, t);\n", "      qx.set_block(mb, t);\n      TL();\n")
    rep("        mma3(oacc[db], lv[(0 * 8 + slot) * 64], lv[(1 * 8 + slot) * 64], ph, pl);\n      }\n    }\n",
        "        mma3(oacc[db], lv[(0 * 8 + slot) * 64], lv[(1 * 8 + slot) * 64], ph, pl);\n      }\n    }\n    TL();\n")
    rep("      ox.set_block(db, t);\n    }\n  }\n", "      ox.set_block(db, t);\n    }\n  }\n  TL();\n")
    rep("    if (active) store_blk<LAT>(x1_out + toff, mb, t, lane);\n", "    if (active) store_blk<LAT>(x1_out + toff, mb, t, lane);\n    TL();\n")
    pre = ('__device__ unsigned long long g_tl[64];\n'
           '#define TL() do { asm volatile("s_nop 0" ::: "memory"); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && tl_n < 64) '
           'g_tl[tl_n] = wall_clock64(); ++tl_n; } while (0)\n')
    post = ('\n}  // namespace gmf\nextern "C" int gmf_dbg_timeline(unsigned long long* out) {\n'
            '  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gmf::g_tl), 64 * sizeof(unsigned long long));\n}\nnamespace gmf {\n')
    text = text[:a] + pre + k + post + text[b:]
    open(path, "w").write(text)
    obj = os.path.join(work, "fusion_wide.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-fno-slp-vectorize", "-c", path, "-o", obj])
    rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o != "fusion_wide.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + rest + ["-o", LIB])
    shutil.rmtree(work)
    print("built", LIB)


def run(M, pe=True):
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    from gmf_amd import _lib
    _lib.LIB_PATH = LIB
    import gmf_amd
    dev = torch.device("cuda:0")
    pio = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                              latent_dim_head=64, pe=pe).to(dev).eval()
    xq = torch.randn(1, M, 256, device=dev)
    img = torch.randn(1, 300, 128, device=dev)
    for _ in range(5):
        pio(img, queries_encoder=xq)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            pio(img, queries_encoder=xq)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "k_fusion_attn_w_h2" in e.key:
            print(f"(this build's kernel: {e.device_time_total / e.count:.1f} us per launch under the torch profiler)")
    lib = _lib.load_library()
    buf = (C.c_ulonglong * 64)()
    lib.gmf_dbg_timeline.restype = C.c_int
    assert lib.gmf_dbg_timeline(buf) == 0
    t = list(buf)[:len(LABELS)]
    print(f"k_fusion_attn_w_h2<PE = {pe}>, M = {M}: wave 0 of workgroup 0, microseconds since its first instruction (10 ns ticks)")
    for i, lab in enumerate(LABELS):
        print(f"  {lab:12s} {(t[i] - t[0]) / 100:8.2f}   (+{(t[i] - t[i - 1]) / 100 if i else 0:6.2f})")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4000, pe=(len(sys.argv) < 4 or sys.argv[3] != 'nope'))
