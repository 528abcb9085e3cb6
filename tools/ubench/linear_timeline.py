"""Where a wave of k_linear_h2 spends its 32-row tile: core-clock stamps (s_memtime) of wave 0 of workgroup (0, 0) at the start, behind
the Q'/K/V projections (12 weight stages), behind the cross-attention (LCPE, LayerNorm, to_q, 7 context tiles, to_out) and behind the
GEGLU feed-forward (48 stages).  Patched COPY of encoder_h2.hip in tools/_ab/libgmf_hip_lin_tl.so.

    python tools/ubench/linear_timeline.py build       # here
    python tools/ubench/linear_timeline.py run B N     # GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
LIB = os.path.join(OUT, "libgmf_hip_lin_tl.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    work = os.path.join(OUT, "src_lin_tl")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip")):
            shutil.copy(os.path.join(CSRC, f), work)
    path = os.path.join(work, "encoder_h2.hip")
    text = open(path).read()

    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) == count, (old[:60], text.count(old))
        text = text.replace(old, new)
    rep("  // ---- Q', K, V from f ---------------------------------------------------------------------------------------------\n",
        "  LTL(0);\n  // ---- Q', K, V from f ---------------------------------------------------------------------------------------------\n")
    rep("  if (PART == 2) ss.prime();\n  // ---- cross-attention: x1 = x' + Wo softmax(q Kc^T) Vc + bo",
        "  LTL(1);\n  if (PART == 2) ss.prime();\n  // ---- cross-attention: x1 = x' + Wo softmax(q Kc^T) Vc + bo")
    rep("  // ---- feed-forward: x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2 --------------------------------------------------------\n",
        "  LTL(2);\n  // ---- feed-forward: x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2 --------------------------------------------------------\n")
    rep("  ff_chunks<NP>(nx, y, lds, ff_wst, lvec_f + 2 * C, lvec_f + 2 * C + FFH, wave, lane, h, 0, FFH / 32);\n#pragma unroll\n  for (int mb = 0; mb < 4; ++mb) {\n    float t[16];",
        "  LTL(3);\n  ff_chunks<NP>(nx, y, lds, ff_wst, lvec_f + 2 * C, lvec_f + 2 * C + FFH, wave, lane, h, 0, FFH / 32);\n  LTL(4);\n#pragma unroll\n  for (int mb = 0; mb < 4; ++mb) {\n    float t[16];")
    # [r4] inside the cross-attention: behind LCPE + LayerNorm + to_q, behind the context tiles, behind to_out
    rep("    f32x16 oacc[2];\n    oacc[0] = zero16(); oacc[1] = zero16();\n    float m_run = -INFINITY, l_half = 0.f;\n    for (int t = 0; t < ttiles; ++t) {\n      const f16x8* lk = as_h2(ss.acquire());",
        "    LTL(5);\n    f32x16 oacc[2];\n    oacc[0] = zero16(); oacc[1] = zero16();\n    float m_run = -INFINITY, l_half = 0.f;\n    for (int t = 0; t < ttiles; ++t) {\n      const f16x8* lk = as_h2(ss.acquire());")
    rep("    FragH2<4> ox;\n    {\n      const float inv = 1.0f / xhalf_sum(l_half);", "    LTL(6);\n    FragH2<4> ox;\n    {\n      const float inv = 1.0f / xhalf_sum(l_half);")
    a = text.index("// k_linear_h2: every linear stage of one encoder layer in ONE pass")
    # [r4] sixteen workgroups over the grid (blockIdx.x in {0, 10, 20, 30} x blockIdx.y in {0, 8, 16, 24}), wave 0 of each; slot 23 of a
    # workgroup's row = the wall clock (s_memrealtime, 100 MHz) at its start, which orders the workgroups in time
    pre = ('__device__ unsigned long long g_ltl[16 * 24];\n'
           '#define LTL(k) do { asm volatile("s_nop 0" ::: "memory"); if (blockIdx.x % 10 == 0 && blockIdx.y % 8 == 0 && threadIdx.x == 0) { '
           'const int wg_ = (blockIdx.y / 8) * 4 + blockIdx.x / 10; g_ltl[wg_ * 24 + (k)] = __builtin_readcyclecounter(); '
           'if ((k) == 0) g_ltl[wg_ * 24 + 23] = __builtin_amdgcn_s_memrealtime(); } asm volatile("s_nop 0" ::: "memory"); } while (0)\n')
    text = text[:a] + pre + text[a:]
    text = text.rstrip()
    assert text.endswith("}  // namespace gmf")
    text = text[:-len("}  // namespace gmf")] + ('}  // namespace gmf\nextern "C" int gmf_dbg_linear_timeline(unsigned long long* out) {\n'
                                               '  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gmf::g_ltl), 16 * 24 * sizeof(unsigned long long));\n}\n')
    open(path, "w").write(text)
    obj = os.path.join(work, "encoder_h2.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-fno-slp-vectorize", "-c", path, "-o", obj])
    rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o != "encoder_h2.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + rest + ["-o", LIB])
    shutil.rmtree(work)
    print("built", LIB)


def run(B, N, pv=1):
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
    _lib.handle_for(0).call("gmf_set_tuning", b"pv_fp8", pv)
    for _ in range(5):
        model(data)
    torch.cuda.synchronize()
    lib = _lib.handle_for(0).lib
    buf = (C.c_ulonglong * (16 * 24))()
    lib.gmf_dbg_linear_timeline.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.gmf_dbg_linear_timeline(buf) == 0
    allv = list(buf)
    rows = [allv[24 * w: 24 * w + 24] for w in range(16)]
    t0 = min(r[23] for r in rows)
    print(f"B={B} N={N} pv_fp8={pv}: k_linear_h2, wave 0 of sixteen workgroups of the last launch; core-clock cycles per phase")
    print("  wg (x, y)   start us |   K/V proj | LCPE+LN+to_q | ctx tiles | to_out | FF seed |     FF |   total")
    for w, v in sorted(enumerate(rows), key=lambda e: e[1][23]):
        print(f"  ({10 * (w % 4):2d}, {8 * (w // 4):2d})   {(v[23] - t0) / 100.0:8.1f} | {v[1] - v[0]:10d} | {v[5] - v[1]:12d} | {v[6] - v[5]:9d} | {v[2] - v[6]:6d} | "
              f"{v[3] - v[2]:7d} | {v[4] - v[3]:6d} | {v[4] - v[0]:7d}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
