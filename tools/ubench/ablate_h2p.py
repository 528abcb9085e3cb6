"""Timing-only ablations of the two per-layer kernels, k_scattn_h2p and k_linear_h2 (results WRONG by construction; never
part of the library).  Each ablation is a set of textual patches applied to a scratch COPY of gmf_amd/csrc; the objects
whose sources changed are rebuilt and linked with the product's other objects into tools/_ab/libgmf_hip_<name>.so
(git-ignored, travels with gpurun).

    python tools/ubench/ablate_h2p.py build            # here (cross-compiles)
    python tools/ubench/ablate_h2p.py run [B] [N]      # GPU box: ms per launch of both kernels for every variant

What each one removes tells what that resource costs the shipped kernel:
    half_lds   K / V fragments re-read from the LDS for every second k-step only (the LDS traffic of a 64-query wave)
    no_c       the compat stream is not fetched (c = 1): HBM / L2 traffic of the cache
    c_l2       the compat tiles come from two alternating addresses (L2 hits): the loads and their waits stay, the HBM stream goes
    no_exp     v_exp_f32 replaced by a subtraction
    no_dma     the in-loop K / V tile refills are not issued (stale tiles)
and for k_linear_h2 (prefix lin_):
    lin_no_gelu     GELU of the feed-forward replaced by the identity
    lin_no_barrier  the stage barriers removed (racy)
    lin_no_exp      the cross-attention's exponentials replaced by a subtraction
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")

EK, EH, FF, MC = "encoder_kernels.hip", "encoder_h2.hip", "enc_ff.hpp", "mfma_core.hpp"
PATCHES = {
    "base": [],
    "half_lds": [
        (EK, "if (pr == 0 && s < 7) { kh_n", "if (pr == 0 && s < 7 && (s & 1)) { kh_n"),
        (EK, "if (pr == 0 && u < 21) {\n", "if (pr == 0 && u < 21 && ((u / 3) & 1)) {\n"),
    ],
    "no_c": [
        (EK, "const f32x4 v = __builtin_nontemporal_load(ct + q * 64);", "const f32x4 v = {1.f, 1.f, 1.f, 1.f}; (void)ct;"),
    ],
    "c_l2": [    # the compat tiles of a wave alternate between TWO addresses: same loads and waits, served by the L2 instead of HBM
        (EK, "    const f32x4* ct = crow + (size_t)t * kCTile16;", "    const f32x4* ct = crow + (size_t)(t & 1) * kCTile16;"),
    ],
    "no_exp": [
        (EK, "        if (u >= 4 && u < 20) x[u - 4] = expo(x[u - 4], m_off);", "        if (u >= 4 && u < 20) x[u - 4] = x[u - 4] - m_off;"),
    ],
    "no_dma": [
        (EK, "        if (u >= 3 && u < 11) issue_piece(t, u - 3);\n", ""),
    ],
    "no_c_no_dma": [    # both streams of the tile loop gone: what is left is the arithmetic (small grids: is the loop latency-bound?)
        (EK, "const f32x4 v = __builtin_nontemporal_load(ct + q * 64);", "const f32x4 v = {1.f, 1.f, 1.f, 1.f}; (void)ct;"),
        (EK, "        if (u >= 3 && u < 11) issue_piece(t, u - 3);\n", ""),
    ],
    # ---- schedule variants of the pv_fp8 form (these compute the SAME results; A/B in one process).  Shipped: K pieces in phase 1
    #      units 0..3, V pieces in phase 2 units 0..3 (1.082 ms); measured against it: all eight in phase 2 1.111, all eight in
    #      phase 1 1.107, K at the tile top 1.131 / 1.124 (same box as 1.124 for the shipped form), V in phase 2 units 4..7 1.135
    # timing-only sensitivities of the pv_fp8 form (wrong results)
    "p8_no_fp8": [         # the four fp8 MFMAs of a tile not issued: 256 matrix-pipe cycles less per tile and wave
        (EK, "          oacc[db] = mfma_f8s(va, pb, oacc[db], db, (int)vsw, psc);\n          if (db < 2)", "          asm volatile(\"\" :: \"v\"(va), \"v\"(pb));\n          if (db < 2)"),
    ],
    "p8_no_hh": [          # the eight P_hi V_hi MFMAs not issued (256 cycles)
        (EK, "          oacc[db] = mfma_h16(vr[u % 3], s2 ? ph1 : ph0, oacc[db]);", "          asm volatile(\"\" :: \"v\"(vr[u % 3]));"),
    ],
    "p8_no_barrier": [     # the tile barrier removed (racy)
        (EK, "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    __syncthreads();\n    if (PVF8) vsw = vsw_next;", "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    if (PVF8) vsw = vsw_next;"),
    ],
    "p8_dma_l2": [         # the refills come from two alternating tiles (L2 / L1 hits): same instructions, same LDS writes, no L2 misses
        (EK, "      if (t + 2 < t_end) dma_piece_1k_s(gk + (size_t)(t + 2) * kStageFloats + (wave + WAVES * q) * 256,",
             "      if (t + 2 < t_end) dma_piece_1k_s(gk + (size_t)(t & 1) * kStageFloats + (wave + WAVES * q) * 256,"),
        (EK, "      dma_piece_1k_s(gv + (size_t)(t + 1) * kStageFloats + (wave + WAVES * (q - 4)) * 256,",
             "      dma_piece_1k_s(gv + (size_t)((t + 1) & 1) * kStageFloats + (wave + WAVES * (q - 4)) * 256,"),
    ],
    "p8_no_softmax": [     # timing only: no exponentials, no splits, no plane conversions - the probabilities' planes are (changing) copies of
                           # the scores' bits; MFMAs, LDS reads, refills, compat product and row maximum stay.  What the softmax's vector work costs.
        (EK, "        if (u >= 4 && u < 20) x[u - 4] = expo(x[u - 4], m_off);", ""),
        (EK, "          if (k < 4) split2h(x[2 * k], x[2 * k + 1], ph0, pl0, 2 * k);\n          else split2h(x[2 * k], x[2 * k + 1], ph1, pl1, 2 * k - 8);",
             "          if (k < 4) { ph0[2 * k] = (_Float16)1.0f; ph0[2 * k + 1] = (_Float16)0.5f; pl0[2 * k] = (_Float16)0.001f; pl0[2 * k + 1] = (_Float16)0.002f; }\n          else { ph1[2 * k - 8] = (_Float16)1.0f; ph1[2 * k - 7] = (_Float16)0.5f; pl1[2 * k - 8] = (_Float16)0.001f; pl1[2 * k - 7] = (_Float16)0.002f; }\n          asm volatile(\"\" :: \"v\"(x[2 * k]), \"v\"(x[2 * k + 1]));"),
        (EK, "        if (u == 20) split2h(x[14], x[15], ph1, pl1, 6);", "        if (u == 20) { ph1[6] = (_Float16)1.0f; ph1[7] = (_Float16)0.5f; pl1[6] = (_Float16)0.001f; pl1[7] = (_Float16)0.002f; asm volatile(\"\" :: \"v\"(x[14]), \"v\"(x[15])); }"),
        (EK, "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n        } else {", "          pb[u] = 0x38383838; ls += 1.0f;\n        } else {"),
    ],
    # ---- the SKELETON of the pv_fp8 tile loop - its matrix stream (24 + 8 f16 MFMAs and 4 block-scaled fp8 MFMAs per
    #      tile and wave), its LDS fragment reads, its LDS-DMA tile ring and its compat stream (four non-temporal 16-byte loads per lane
    #      and tile, consumed one tile later) with NO softmax: no compat product, no row maximum, no exponentials, no splits, no plane
    #      conversions, no row sums, no rescale.  The probabilities' operands are the scores' own bits (8 v_cvt_pk_f16_f32 per tile keep the
    #      data changing: an MFMA stream on constant operands runs at a higher clock and proves nothing).  What pipe-GHz does the chip
    #      sustain with everything the kernel streams but none of its vector work?  p8_skel keeps the tile barrier, p8_skel_nobar drops it
    #      (racy), p8_skel_nostream also drops the compat loads and the ring refills (the bare matrix + LDS-read stream of this kernel).
    "p8_skel": "SKEL",
    "p8_skel_nobar": "SKEL+NOBAR",
    "p8_skel_nostream": "SKEL+NOBAR+NOSTREAM",
    "p8_skel_noc": "SKEL+NOBAR+NOC",              # ... no compat loads, ring refills kept
    "p8_skel_norefill": "SKEL+NOBAR+NOREFILL",    # ... no ring refills (stale K / V tiles), compat loads kept
    # [r5] timing only: what would the tile cost with the two CROSS products of S = K Q'^T on the block-scaled fp8 pipe as well (8 f16 +
    #      4 fp8 MFMAs instead of 24 f16 in phase 1: 512 instead of 768 matrix-pipe cycles; same LDS reads - the low plane's bytes stand in
    #      for the e4m3 planes -, same vector work in the gaps)?  Results are garbage; the question is the microseconds.
    "p8_qk8_timing": [
        (EK, "        s_next = mma3_part(pr, s_next, kh, kl, qh[s], ql[s]);\n        if (pr == 2) { kh = kh_n; kl = kl_n; }\n        if (u < 4) {",
             "        if (pr == 2) s_next = mfma_h16(kh, qh[s], s_next);\n"
             "        else if (pr == 0 && (s & 1)) {\n"
             "          const i32x4 a0 = __builtin_bit_cast(i32x4, kl), a1 = __builtin_bit_cast(i32x4, kl_n);\n"
             "          const i32x4 b0 = __builtin_bit_cast(i32x4, ql[s]), b1 = __builtin_bit_cast(i32x4, ql[s - 1]);\n"
             "          const i32x8 a8 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, b8 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};\n"
             "          s_next = mfma_f8s(a8 & 0x7e7e7e7e, b8 & 0x7e7e7e7e, s_next, 0, 0x6b6b6b6b, 0x6b6b6b6b);   // (no NaN bytes, scales 2^-20: S stays the hi x hi product - realistic data everywhere else)\n"
             "        }\n        if (pr == 2) { kh = kh_n; kl = kl_n; }\n        if (u < 4) {"),
    ],
    # [r5] schedule variants of the refills after the cross products of S moved to the fp8 pipe (the tile is 256 matrix-pipe cycles shorter, so
    #      every lead time measured in units shrank): K_{t+2} / V_{t+1} issued at other units of phase 1.  Same results as the shipped kernel.
    "r5_dma_6_14": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 6) issue_k4(t);\n        if (u == 14) issue_v4(t);\n")],
    "r5_dma_4_10": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 4) issue_k4(t);\n        if (u == 10) issue_v4(t);\n")],
    "r5_dma_0_8": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 0) issue_k4(t);\n        if (u == 8) issue_v4(t);\n")],
    "r5_dma_8_16": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 8) issue_k4(t);\n        if (u == 16) issue_v4(t);\n")],
    "r5_dma_v_first": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 4) issue_v4(t);\n        if (u == 14) issue_k4(t);\n")],
    # [r5, late] the wait at the tile top (attn_timeline.py: 458 of a wave's ~4200 cycles per tile) covers the V pieces issued at unit 23 of
    # the previous tile, which nothing reads before unit 21: timing-only (the counts are wrong in a workgroup's last tiles)
    "r5_vm4": [(EK, '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    vsw = vsw_next;\n',
                '    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");\n    __syncthreads();\n    vsw = vsw_next;\n')],
    "r5_vm4_mid": [(EK, '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    vsw = vsw_next;\n',
                    '    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");\n    __syncthreads();\n    vsw = vsw_next;\n'),
                   (EK, "        if (u >= 21) vr[u - 21] = lv[hslot(u - 21)];\n        __builtin_amdgcn_sched_barrier(0);\n",
                    '        if (u == 21) { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); __syncthreads(); }\n        if (u >= 21) vr[u - 21] = lv[hslot(u - 21)];\n        __builtin_amdgcn_sched_barrier(0);\n')],
    # the compat cache written with non-temporal stores (3.2 GB per forward at 32 x 5000, re-read by 12 streaming launches)
    "r5_c_nt_store": [("enc_common.hpp", "      for (int q = 0; q < 4; ++q) ct[q * 64] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);",
                       "      for (int q = 0; q < 4; ++q) { const f32x4 w = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}; __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(ct + q * 64)); }")],
    "r5_dma_16_23": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n")],
    "r5_dma_14_22": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 14) issue_k4(t);\n        if (u == 22) issue_v4(t);\n")],
    "r5_dma_12_p2": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 12) issue_k4(t);\n"), (EK, "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n", "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n" "          if (u == 2) issue_v4(t);\n")],
    "r5_dma_20_p2": [(EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u == 20) issue_k4(t);\n"), (EK, "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n", "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n" "          if (u == 4) issue_v4(t);\n")],
    "p8_no_consist": [     # timing only: the row sum without the decoded low plane (what the consistent sum costs: nothing measurable)
        (EK, "      ls_l = __builtin_amdgcn_fdot2(dec2_fp8_f16<true>(pb[4 + w], 0x1p-10f), ones, ls_l, false);", ""),
        (EK, "      ls_l = __builtin_amdgcn_fdot2(dec2_fp8_f16<false>(pb[4 + w], 0x1p-10f), ones, ls_l, false);", ""),
    ],
    "lin_no_qkv_stores": [ # timing only: the Q' / K / V images are not stored (is the projection phase waiting for its own stores through the in-order vmcnt?)
        (EH, "        store_block_h2(dst, mb, t, lane);\n      }\n    }\n    unsigned vsw = 0;", "        if (lane > 100000) store_block_h2(dst, mb, t, lane);\n        asm volatile(\"\" :: \"v\"(t[0]), \"v\"(t[5]), \"v\"(t[10]), \"v\"(t[15]));\n      }\n    }\n    unsigned vsw = 0;"),
        (EH, "      if (v_scale) store_block_v8(v_out + toff, db, t, lane, vsw);     // (4 stores of 16 bytes either way: the counted waits hold)\n      else store_block_h2(v_out + toff, db, t, lane);",
             "      if (lane > 100000) store_block_h2(v_out + toff, db, t, lane);\n      asm volatile(\"\" :: \"v\"(t[0]), \"v\"(t[5]), \"v\"(t[10]), \"v\"(t[15]));"),
    ],
    "p8_dma_pieces": [     # the attention tile ring refilled by eight separately set-up pieces (K at units 12..15, V at 20..23) instead of two statements
        (EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", "        if (u >= 12 && u < 16) issue_piece(t, u - 12);\n        if (u >= 20) issue_piece(t, u - 16);\n"),
        (EK, "        if (PVF8) { issue_k4(t); issue_v4(t); }", "        if (false) { }"),
    ],
    "lin_dma_pieces": [    # the stage refills as four separately set-up 1 KiB pieces per wave (the form before dma_4k_s)
        (MC, "      dma_4k_s(g + wave * 1024, dst + wave * 1024, (unsigned)lane * 16u);\n      ++issued;",
             "#pragma unroll\n      for (int q = 0; q < 4; ++q) dma_piece_1k_s(g + (wave + 4 * q) * 256, dst + (wave + 4 * q) * 256, (unsigned)lane * 16u);\n      ++issued;"),
        (FF, "    dma_4k_s(g + wave * 1024, dst + wave * 1024, (unsigned)lane * 16u);   // (statement form, one M0 setup per stage: see StageRing)",
             "#pragma unroll\n    for (int q = 0; q < 4; ++q) dma_piece_1k_s(g + (wave + 4 * q) * 256, dst + (wave + 4 * q) * 256, (unsigned)lane * 16u);"),
    ],
    "lin_no_gelu": [
        (FF, "a_cur[u] *= gelu_erf_1r(g_cur[u]);", "a_cur[u] *= g_cur[u];"),
    ],
    "lin_no_barrier": [
        (MC, "    asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(YOUNGER) : \"memory\");\n    __syncthreads();", "    asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(YOUNGER) : \"memory\");"),
        (MC, "    __syncthreads();            // everyone's pieces of this stage landed; the slot of the previous stage is free\n", ""),
        (FF, "    asm volatile(\"s_waitcnt vmcnt(8)\" ::: \"memory\");   // all but this wave's pieces of the 2 younger stages have landed\n    __syncthreads();",
         "    asm volatile(\"s_waitcnt vmcnt(8)\" ::: \"memory\");"),
    ],
    "lin_no_exp": [
        (EH, "      for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }\n      l_half = fmaf(l_half, alpha, ls);\n      if (__any(moved)) {\n#pragma unroll\n        for (int db = 0; db < 2; ++db)",
         "      for (int r = 0; r < 16; ++r) { x[r] = x[r] - m_off; ls += x[r]; }\n      l_half = fmaf(l_half, alpha, ls);\n      if (__any(moved)) {\n#pragma unroll\n        for (int db = 0; db < 2; ++db)"),
    ],
}

_SKEL = [
    (EK, """        if (u < 4) {
#pragma unroll
          for (int r = 4 * u; r < 4 * u + 4; r += 2) {
            x[r] = score(r, s_cur[r]);
            x[r + 1] = score(r + 1, s_cur[r + 1]);
            mx = __builtin_fmaxf(mx, __builtin_fmaxf(x[r], x[r + 1]));
          }
        }
""", """        if (u == 0) asm volatile("" :: "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
"""),
    (EK, """          mx = xhalf_max_swap(mx);
          if (CFMT == 2) mx *= kInvU16;
          // LAZY reference""", """          if (false) {
          // LAZY reference"""),
    (EK, """          m_off = m_new - (10.0f - kLazy);
          rescale(moved, alpha);
        }
        if (u >= 4 && u < 20) x[u - 4] = expo(x[u - 4], m_off);""", """          }
        }"""),
    (EK, """          const bool moved = mx > m_run + kLazy;
          const float m_new = moved ? mx : m_run;
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
""", ""),
    (EK, """          if (k < 4) split2h(x[2 * k], x[2 * k + 1], ph0, pl0, 2 * k);
          else split2h(x[2 * k], x[2 * k + 1], ph1, pl1, 2 * k - 8);""",
     """          { const f32x2 xx_ = {s_cur[2 * k], s_cur[2 * k + 1]}; const f16x2 hh_ = __builtin_convertvector(xx_, f16x2);
            if (k < 4) { ph0[2 * k] = hh_[0]; ph0[2 * k + 1] = hh_[1]; pl0[2 * k] = hh_[1]; pl0[2 * k + 1] = hh_[0]; }
            else { ph1[2 * k - 8] = hh_[0]; ph1[2 * k - 7] = hh_[1]; pl1[2 * k - 8] = hh_[1]; pl1[2 * k - 7] = hh_[0]; } }"""),
    (EK, "        if (u == 20) split2h(x[14], x[15], ph1, pl1, 6);",
     "        if (u == 20) { const f32x2 xx_ = {s_cur[14], s_cur[15]}; const f16x2 hh_ = __builtin_convertvector(xx_, f16x2); ph1[6] = hh_[0]; ph1[7] = hh_[1]; pl1[6] = hh_[1]; pl1[7] = hh_[0]; }"),
    (EK, "          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);\n        } else {",
     "          pb[u] = __builtin_bit_cast(i32x4, u < 4 ? ph0 : pl1)[u & 3] & 0x3f3f3f3f; ls += 1.0f;\n        } else {"),
]
_NOBAR = [
    (EK, "    float mx = -INFINITY, m_off = 0.f, alpha = 1.f, ls = 0.f, ls_l = 0.f;\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    __syncthreads();\n    vsw = vsw_next;",
     "    float mx = -INFINITY, m_off = 0.f, alpha = 1.f, ls = 0.f, ls_l = 0.f;\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    vsw = vsw_next;"),
]
_NOC = [
    (EK, "const f32x4 v = __builtin_nontemporal_load(ct + q * 64);", "const f32x4 v = {1.f, 1.f, 1.f, 1.f}; (void)ct;"),
    (EK, "    if (PVF8) vsw_next = __builtin_nontemporal_load(vsrow + (size_t)t * 64);", "    if (PVF8) vsw_next = 0x7f7f7f7fu;"),
]
_NOREFILL = [
    (EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", ""),
]
_NOSTREAM = [
    (EK, "const f32x4 v = __builtin_nontemporal_load(ct + q * 64);", "const f32x4 v = {1.f, 1.f, 1.f, 1.f}; (void)ct;"),
    (EK, "        if (u == 16) issue_k4(t);\n        if (u == 23) issue_v4(t);\n", ""),
    (EK, "    if (PVF8) vsw_next = __builtin_nontemporal_load(vsrow + (size_t)t * 64);", "    if (PVF8) vsw_next = 0x7f7f7f7fu;"),
]
for _k, _v in list(PATCHES.items()):
    if isinstance(_v, str):
        PATCHES[_k] = sum(({"SKEL": _SKEL, "NOBAR": _NOBAR, "NOSTREAM": _NOSTREAM, "NOC": _NOC, "NOREFILL": _NOREFILL}[t] for t in _v.split("+")), [])
OBJ_OF = {EK: ["encoder_kernels"], EH: ["encoder_h2"], FF: ["encoder_kernels", "encoder_h2"], MC: ["encoder_kernels", "encoder_h2"],
          "enc_common.hpp": ["encoder_kernels", "encoder_h2"]}


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    for name, patches in PATCHES.items():
        if only and name not in only:
            continue
        work = os.path.join(OUT, "src_" + name)
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        for f in os.listdir(CSRC):
            if f.endswith((".hpp", ".hip")):
                shutil.copy(os.path.join(CSRC, f), work)
        rebuild = set() if patches else {"encoder_kernels"}
        for fname, old, new in patches:
            path = os.path.join(work, fname)
            text = open(path).read()
            if text.count(old) < 1:
                raise SystemExit(f"{name}: pattern not found in {fname}: {old[:50]!r}")
            open(path, "w").write(text.replace(old, new))
            rebuild.update(OBJ_OF[fname])
        objs = []
        for o in sorted(rebuild):
            obj = os.path.join(work, o + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                                   "-fno-slp-vectorize", "-c", os.path.join(work, o + ".hip"), "-o", obj])
            objs.append(obj)
        rest = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o[:-2] not in rebuild]
        lib = os.path.join(OUT, f"libgmf_hip_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + rest + ["-o", lib])
        shutil.rmtree(work)
        print("built", lib, flush=True)


def run(argv):
    names = [a for a in argv if a in PATCHES] or list(PATCHES)
    argv = [a for a in argv if a not in PATCHES]
    for name in names:
        lib = os.path.join(OUT, f"libgmf_hip_{name}.so")
        env = dict(os.environ, GMF_LIB=lib, ROWS=os.environ.get("ROWS", "3"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_times.py")] + argv, env=env,
                           capture_output=True, text=True)
        print(f"== {name}", flush=True)
        print("\n".join(l for l in r.stdout.splitlines() if "scattn" in l or "k_linear" in l or "k_small" in l or "compat_build" in l or "encode:" in l), flush=True)
        if r.returncode:
            print(r.stderr[-1500:])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:] or None)
    else:
        run(sys.argv[2:])
