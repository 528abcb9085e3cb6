"""Per-stage stamps of k_linear_h2's first phase (the Q' / K / V projections: 12 weight stages of 16 KiB, 24 MFMAs each) - VERDICT r3
item 2 (i): "find what a V stage waits for by stamping each s_waitcnt".  Wave 0 of FOUR workgroups of the grid (first, two in the
middle, last: a workgroup in the middle of the launch has a partner workgroup in another phase on its CU) takes s_memtime stamps

    s0  in front of the counted vmcnt wait of the stage's acquire
    s1  behind that wait                        (s1 - s0: waiting for this wave's own DMA pieces / older stores)
    s2  behind the workgroup barrier + the refill issue    (s2 - s1: waiting for the other three waves)
    s3  behind the stage's 24 MFMAs             (768 matrix-pipe cycles)
    s4  behind bias + split + the 4 stores      (Q' / K: fp16 planes; V: fp16 high plane + two e4m3 planes + the tile maximum)

The stamps go to a 2 KiB tail of the kernel's own LDS array by ds_write (NOT to global memory: a global store would join the vmcnt
queue the counted waits count) and leave through a debug buffer at the end.  Patched COPIES of mfma_core.hpp / encoder_h2.hip in
tools/_ab/libgmf_hip_lin_st.so - never the library.  Read SHARES, not lengths (every stamp drains lgkmcnt).

    python tools/ubench/linear_stage_stamps.py build       # here
    python tools/ubench/linear_stage_stamps.py run B N     # GPU box
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
LIB = os.path.join(OUT, "libgmf_hip_lin_st.so")
NST, NK, NWG = 30, 6, 4      # stages stamped (12 + 2 + ttiles + 2 fit), stamps per stage, workgroups (LDS: 77 KiB + 1.25 KiB, two workgroups per CU still fit)


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    work = os.path.join(OUT, "src_lin_st")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip")):
            shutil.copy(os.path.join(CSRC, f), work)

    # ---- the ring: stamps around the wait, the barrier and the refill of acquire_counted / acquire --------------------------
    path = os.path.join(work, "mfma_core.hpp")
    text = open(path).read()

    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) == count, (old[:70], text.count(old))
        for tok, val in (("@NST@", NST), ("@NK@", NK), ("@PER@", NST * NK), ("@TOT@", NWG * NST * NK), ("@LDSF@", 2 * NST * NK)):
            new = new.replace(tok, str(val))
        text = text.replace(old, new)
    rep("  int issued, consumed, total;\n  int wave, lane;\n",
        "  int issued, consumed, total;\n  int wave, lane;\n  unsigned long long* st_buf;   // stamp area (LDS) or null\n"
        "  GMF_DEVINL void stamp(int stage, int k) {\n"
        "    __builtin_amdgcn_sched_barrier(0);\n"
        "    unsigned long long t_;\n"
        "    asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\");\n"
        "    if (st_buf && lane == 0 && stage < @NST@) st_buf[stage * @NK@ + k] = t_;\n"
        "    __builtin_amdgcn_sched_barrier(0);\n  }\n")
    rep("    base = lds_base; issued = 0; consumed = 0;\n", "    base = lds_base; issued = 0; consumed = 0; st_buf = nullptr;\n")
    rep("    asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(YOUNGER) : \"memory\");\n    __syncthreads();\n    const float* cur = base + (consumed % NBUF) * kStageFloats;\n    ++consumed;\n    issue_one();\n",
        "    stamp(consumed, 0);\n    asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(YOUNGER) : \"memory\");\n    stamp(consumed, 1);\n    __syncthreads();\n"
        "    const float* cur = base + (consumed % NBUF) * kStageFloats;\n    ++consumed;\n    issue_one();\n    stamp(consumed - 1, 2);\n")
    rep("    if (issued - consumed == NBUF - 1) asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(4 * (NBUF - 2)) : \"memory\");\n    else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    __syncthreads();            // everyone's pieces of this stage landed; the slot of the previous stage is free\n    const float* cur = base + (consumed % NBUF) * kStageFloats;\n    ++consumed;\n    issue_one();\n",
        "    stamp(consumed, 0);\n    if (issued - consumed == NBUF - 1) asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(4 * (NBUF - 2)) : \"memory\");\n    else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    stamp(consumed, 1);\n    __syncthreads();\n"
        "    const float* cur = base + (consumed % NBUF) * kStageFloats;\n    ++consumed;\n    issue_one();\n    stamp(consumed - 1, 2);\n")
    open(path, "w").write(text)

    # ---- the kernel: stamp area behind its LDS, stamps behind the MFMAs and behind the stores of the 12 stages ---------------
    path = os.path.join(work, "encoder_h2.hip")
    text = open(path).read()
    rep("constexpr int kLinLdsFloats = 4 * kStageFloats + 7 * C + (3 * C + 2 * FFH) + kWavesPerWG * 2 * C;",
        "constexpr int kLinLdsFloatsOrig = 4 * kStageFloats + 7 * C + (3 * C + 2 * FFH) + kWavesPerWG * 2 * C;\n"
        "constexpr int kLinLdsFloats = kLinLdsFloatsOrig + @LDSF@;\n"
        "__device__ unsigned long long g_lst[@TOT@ + 16];\n"
        "__device__ __forceinline__ int lst_slot(int T_) {   // which of the stamped workgroups this one is (-1: none)\n"
        "  const int gx = gridDim.x, gy = gridDim.y, x = blockIdx.x, y = blockIdx.y;\n"
        "  if (x == 0 && y == 0) return 0;\n  if (x == gx / 2 && y == gy / 3) return 1;\n"
        "  if (x == gx / 3 && y == (2 * gy) / 3) return 2;\n  if (x == gx - 1 && y == gy - 1) return 3;\n  return -1;\n}\n")
    rep("  StageRing<4> ss;\n  if (PART == 0) ss.init(",
        "  StageRing<4> ss;\n  const int lst_wg = PART == 0 ? lst_slot(T) : -1;\n"
        "  unsigned long long* const lst = reinterpret_cast<unsigned long long*>(lds + kLinLdsFloatsOrig);\n"
        "  if (PART == 0 && lst_wg >= 0 && wave == 0) { for (int q = lane; q < @PER@; q += 64) lst[q] = 0ull; }\n"
        "  if (PART == 0) ss.init(")
    rep("    ss.prime();                                  // AFTER the loads above",
        "    if (PART == 0 && lst_wg >= 0 && wave == 0) { ss.st_buf = lst; unsigned long long t0_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t0_) :: \"memory\"); if (lane == 0) g_lst[@TOT@ + lst_wg] = t0_; }\n"
        "    ss.prime();                                  // AFTER the loads above")
    rep("        mma_wx_h2n<8, NP>(acc, lw, fx);\n        float t[16];\n#pragma unroll\n        for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bqk[which][16 * mb + r]);\n        store_block_h2(dst, mb, t, lane);\n",
        "        mma_wx_h2n<8, NP>(acc, lw, fx);\n        ss.stamp(ss.consumed - 1, 3);\n        float t[16];\n#pragma unroll\n        for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bqk[which][16 * mb + r]);\n"
        "        { f16x8 hi_[2], lo_[2];\n          split8h(&t[0], hi_[0], lo_[0]);\n          split8h(&t[8], hi_[1], lo_[1]);\n          ss.stamp(ss.consumed - 1, 4);\n"
        "          f16x8* b_ = reinterpret_cast<f16x8*>(dst);\n"
        "          for (int half = 0; half < 2; ++half) { b_[(0 * 8 + 2 * mb + half) * 64 + lane] = hi_[half]; b_[(1 * 8 + 2 * mb + half) * 64 + lane] = lo_[half]; } }\n"
        "        ss.stamp(ss.consumed - 1, 5);\n")
    rep("      mma_xw_h2n<8, NP>(acc, lw, fx);\n      float t[16];\n#pragma unroll\n      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bvv[db]);\n      if (v_scale) store_block_v8(v_out + toff, db, t, lane, vsw);     // (4 stores of 16 bytes either way: the counted waits hold)\n      else store_block_h2(v_out + toff, db, t, lane);\n",
        "      mma_xw_h2n<8, NP>(acc, lw, fx);\n      ss.stamp(ss.consumed - 1, 3);\n      float t[16];\n#pragma unroll\n      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bvv[db]);\n      if (v_scale) store_block_v8(v_out + toff, db, t, lane, vsw);     // (4 stores of 16 bytes either way: the counted waits hold)\n      else store_block_h2(v_out + toff, db, t, lane);\n      ss.stamp(ss.consumed - 1, 5);\n")
    rep("      float xp[CF];\n      load_frag_p32<CF>(xp, f_in + toff, lane);\n      if (PART == 2) {",
        "      float xp[CF];\n      ss.stamp(28, 0);\n      load_frag_p32<CF>(xp, f_in + toff, lane);\n      if (PART == 0) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); ss.stamp(28, 1); }\n      if (PART == 2) {")
    rep("      LcpeHalo<CF>::apply(xp, halo, lvec_a, tile * 32 + i, N, lane);\n", "      LcpeHalo<CF>::apply(xp, halo, lvec_a, tile * 32 + i, N, lane);\n      ss.stamp(28, 2);\n")
    rep("          layernorm_frag<CF>(xn, xp, lvec_a + 4 * C, lvec_a + 5 * C, h);\n          nx.set(xn);\n        }\n", "          layernorm_frag<CF>(xn, xp, lvec_a + 4 * C, lvec_a + 5 * C, h);\n          nx.set(xn);\n        }\n        ss.stamp(28, 3);\n")
    # the stamps leave through g_lst when the first phase is over (the ring keeps stamping the later stages' acquires until then)
    rep("  // ---- feed-forward: x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2 --------------------------------------------------------\n",
        "  if (PART == 0 && lst_wg >= 0 && wave == 0) {\n    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
        "    for (int q = lane; q < @PER@; q += 64) g_lst[lst_wg * @PER@ + q] = lst[q];\n    ss.st_buf = nullptr;\n  }\n"
        "  // ---- feed-forward: x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2 --------------------------------------------------------\n")
    text = text.rstrip()
    assert text.endswith("}  // namespace gmf")
    tail = ('}  // namespace gmf\nextern "C" int gmf_dbg_linear_stamps(unsigned long long* out) {\n'
                                               '  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gmf::g_lst), (@TOT@ + 16) * sizeof(unsigned long long));\n}\n').replace("@TOT@", str(NWG * NST * NK))
    text = text[:-len("}  // namespace gmf")] + tail
    open(path, "w").write(text)
    objs = []
    for name in ("encoder_h2", "encoder_kernels"):      # both include mfma_core.hpp; only encoder_h2 needs the stamps, the ring's layout must agree
        obj = os.path.join(work, name + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-fno-slp-vectorize", "-c", os.path.join(work, name + ".hip"), "-o", obj])
        objs.append(obj)
    rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o not in ("encoder_h2.o", "encoder_kernels.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + rest + ["-o", LIB])
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
    for _ in range(8):
        model(data)
    torch.cuda.synchronize()
    lib = _lib.handle_for(0).lib
    n = NWG * NST * NK + 16
    buf = (C.c_ulonglong * n)()
    lib.gmf_dbg_linear_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.gmf_dbg_linear_stamps(buf) == 0
    v = list(buf)
    names = [f"Q' {i}" for i in range(4)] + [f"K {i}" for i in range(4)] + [f"V {i}" for i in range(4)]
    print(f"B={B} N={N} pv_fp8={pv}: k_linear_h2 first phase, wave 0 of four workgroups of the LAST launch (layer 11); core-clock cycles")
    print("  per stage: vmcnt wait | barrier + refill issue | 24 MFMAs (768 pipe cycles) | bias + split (Q'/K only) | 4 stores issued | whole stage")
    for wg in range(NWG):
        s = [[v[(wg * NST + st) * NK + k] for k in range(NK)] for st in range(NST)]
        t0 = v[NWG * NST * NK + wg]
        if not s[0][0]:
            print(f"  workgroup slot {wg}: not stamped")
            continue
        print(f"  workgroup slot {wg}: prime -> first acquire {s[0][0] - t0}")
        for st in range(12):
            a = s[st]
            nxt = s[st + 1][0] if s[st + 1][0] else a[5]
            mid = a[4] if a[4] else a[3]
            print(f"    {names[st]:5s} {a[1] - a[0]:6d} | {a[2] - a[1]:6d} | {a[3] - a[2]:6d} | {mid - a[3]:6d} | {a[5] - mid:6d} | {nxt - a[0]:6d}")
        q = s[28]
        print(f"    phase-2 prologue: V 3 stores issued -> f reload issued {q[0] - s[11][5]}; reload landed (vmcnt(0): every older store acknowledged) {q[1] - q[0]}; "
              f"LCPE {q[2] - q[1]}; LayerNorm + split {q[3] - q[2]}; -> first to_q acquire {s[12][0] - q[3]}")
        print(f"    phase: {s[12][0] - s[0][0]} cycles for 12 stages; later acquires (to_q 2, context tiles, to_out 2): wait | barrier")
        print("      " + "  ".join(f"{s[st][1] - s[st][0]}|{s[st][2] - s[st][1]}" for st in range(12, 12 + 2 + 7 + 2) if s[st][0]))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
