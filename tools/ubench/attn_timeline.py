"""Where a wave of the attention kernel (pv_fp8 form, tile_step_f8) spends a key tile: core-clock stamps (s_memtime) taken by wave 0
of workgroup 0 at the entry of a tile step, behind the tile barrier, behind phase 1 (S_{t+1} + the softmax of tile t) and behind
phase 2 (O += P V).  The stamps live in a patched COPY of encoder_kernels.hip built into tools/_ab/libgmf_hip_attn_tl.so.

    python tools/ubench/attn_timeline.py build          # here (cross-compiles)
    python tools/ubench/attn_timeline.py run B N        # GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
LIB = os.path.join(OUT, "libgmf_hip_attn_tl.so")
NT = 24          # tiles recorded
NS = 10          # stamps per tile


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    work = os.path.join(OUT, "src_attn_tl")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip")):
            shutil.copy(os.path.join(CSRC, f), work)
    path = os.path.join(work, "encoder_kernels.hip")
    text = open(path).read()

    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) == count, (old[:60], text.count(old))
        text = text.replace(old, new)
    # stamps: 4 per tile step
    rep("  auto tile_step_f8 = [&](const int t, const f32x16& s_cur, f32x16& s_next) {\n    float x[16];\n",
        "  auto tile_step_f8 = [&](const int t, const f32x16& s_cur, f32x16& s_next) {\n    TL(t, 0);\n    float x[16];\n")
    rep('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    vsw = vsw_next;\n',
        '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    TL(t, 9);\n    __syncthreads();\n    TL(t, 1);\n    vsw = vsw_next;\n')
    rep("        if (u >= 21) vr[u - 21] = lv[hslot(u - 21)];\n        __builtin_amdgcn_sched_barrier(0);\n      }\n    }\n",
        "        if (u >= 21) vr[u - 21] = lv[hslot(u - 21)];\n        __builtin_amdgcn_sched_barrier(0);\n"
        "        if (u == 3) TL(t, 2); if (u == 7) TL(t, 3); if (u == 11) TL(t, 4); if (u == 19) TL(t, 5);\n      }\n    }\n    TL(t, 6);\n")
    rep("          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n        } else {",
        "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n          if (u == 7) TL(t, 7);\n        } else {")
    rep("    l_half = fmaf(l_half, alpha, ls + ls_l);\n  };\n", "    TL(t, 8);\n    l_half = fmaf(l_half, alpha, ls + ls_l);\n  };\n")
    a = text.index("// PVF8 (with NPROD = 3): the two CROSS products")
    pre = (f'__device__ unsigned long long g_tl[{NT * NS}];\n'
           '#define TL(t, k) do { asm volatile("s_nop 0" ::: "memory"); if (bid == ' + os.environ.get('TL_BID', '0') + ' && threadIdx.x == 0 && (t) - t_begin < ' + str(NT) + ') '
           'g_tl[((t) - t_begin) * ' + str(NS) + ' + (k)] = __builtin_readcyclecounter(); asm volatile("s_nop 0" ::: "memory"); } while (0)\n')
    text = text[:a] + pre + text[a:]
    text = text.rstrip()
    assert text.endswith("}  // namespace gmf")
    text = text[:-len("}  // namespace gmf")] + ('}  // namespace gmf\nextern "C" int gmf_dbg_attn_timeline(unsigned long long* out) {\n'
                                               f'  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gmf::g_tl), {NT * NS} * sizeof(unsigned long long));\n}}\n')
    open(path, "w").write(text)
    obj = os.path.join(work, "encoder_kernels.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-fno-slp-vectorize", "-c", path, "-o", obj])
    rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o != "encoder_kernels.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + rest + ["-o", LIB])
    shutil.rmtree(work)
    print("built", LIB)


def run(B, N):
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    from gmf_amd import _lib
    _lib.LIB_PATH = LIB
    import gmf_amd
    from gmf_amd import synthetic
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
    lib = _lib.handle_for(0).lib
    buf = (C.c_ulonglong * (NT * NS))()
    lib.gmf_dbg_attn_timeline.argtypes = [C.POINTER(C.c_ulonglong)]
    rc = lib.gmf_dbg_attn_timeline(buf)
    assert rc == 0, rc
    v = list(buf)
    names = ["vmcnt(0) + barrier wait", "u0-3 scores+max, fetch c, offset", "u4-7", "u8-11", "u12-19 (+K pieces at 16)", "u20-23 (+V pieces at 23)", "ph2 hh u0-7 (+conversions)", "ph2 fp8 u8-11", "to next entry"]
    print(f"B={B} N={N}: wave 0 of workgroup 0, last launch; core-clock cycles per key tile")
    tot = [0] * 9
    n = 0
    vm = 0
    for t in range(1, NT - 1):
        st = v[NS * t: NS * t + 9]
        nxt = v[NS * (t + 1)]
        if not (st[0] and nxt):
            break
        row = [st[k + 1] - st[k] for k in range(8)] + [nxt - st[8]]
        tot = [a + b_ for a, b_ in zip(tot, row)]
        vm += v[NS * t + 9] - st[0]
        n += 1
    if n:
        for nm, x in zip(names, tot):
            print(f"  {nm:34s} {x / n:8.0f}")
        print(f"  {'(of the first line: the vmcnt(0) wait)':34s} {vm / n:8.0f}")
        print(f"  {'tile':34s} {sum(tot) / n:8.0f}   (matrix-pipe time of the tile: 16 x 32 + 8 x 64 = 1024; phase 1 holds 8 x 32 + 4 x 64 = 512 of it, phase 2 the same; the stamps themselves stretch the tile by ~10 %: more of them distort it beyond use)")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]), int(sys.argv[3]))
