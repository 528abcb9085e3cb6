"""CPU emulation (checker-side tool, never the product): what would the spatial-consistency attention's logits lose if the two CROSS
products of the split-fp16 scheme (hi x lo, lo x hi) ran on the block-scaled fp8 / fp6 MFMA instead of the f16 MFMA?

The whole encoder runs in fp64 (oracle functions on fp64 tensors) except the two contractions of `sc_attention`, which are
replaced by an emulation of the kernel's operand arithmetic:
  split3   q.k = qh.kh + qh.kl + ql.kh, all planes fp16                                  (what k_scattn_h2p multiplies today)
  fp8x     q.k = qh.kh + e4m3(q).e4m3s(kl) + e4m3s(ql).e4m3(k)   (cross terms on v_mfma_scale_f32_32x32x64_f8f6f4, 2x rate)
  fp6x     the same with e2m3 cross operands (4x rate)
  one      q.k = qh.kh                                                                    (the throughput mode)
and the same for P.V.  The deviation of the logits from the all-fp64 evaluation is printed per scheme, next to the fp32
oracle's own deviation.     python tests/tools/mx_cross_emulation.py [kind] [N] [seeds...] [--cond] [--qk] [--guard]
  --guard [r5]: the schemes the library ships since round 5 - `fp8pv_g` (cross products of P V on e4m3) and `fp8x_g` (those of
  Q' K^T as well, one scale per (row, 32-channel block)) in the layers the device-side guard lets through (score bound
  (|Wq|_2 F + |bq|)(|Wk|_2 F + |bk|) / sqrt(C) <= 1024, gmf_pack.cpp pv_guard_threshold), three f16 products in the others; the
  number of guarded layers is printed in brackets."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gmf_amd import synthetic
from oracle import gmf_oracle as O

torch.set_num_threads(8)


def f16(x):
    return x.float().to(torch.float16).double()


def _pow2_floor(x):
    return torch.exp2(torch.floor(torch.log2(x.clamp_min(1e-300))))


def e4m3(x, dim):
    """block-scaled OCP e4m3: one power-of-two scale per block along `dim`, the block's maximum lands in [128, 256)."""
    scale = _pow2_floor(x.abs().amax(dim=dim, keepdim=True)) / 128.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    return (x / scale).float().to(torch.float8_e4m3fn).double() * scale


def e2m3(x, dim):
    """block-scaled fp6 e2m3 (max 7.5, 3 mantissa bits, subnormal step 1/8); block maximum in [4, 8) (saturates at 7.5)."""
    scale = _pow2_floor(x.abs().amax(dim=dim, keepdim=True)) / 4.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    y = x / scale
    a = y.abs()
    e = torch.floor(torch.log2(a.clamp_min(1.0)))            # binade of normals (>= 1), subnormals share the step of [1, 2)
    step = torch.exp2(e - 3)
    r = torch.round(a / step) * step
    return torch.sign(y) * r.clamp_max(7.5) * scale


def tiles(x, dim):      # pad dim to a multiple of 32 and view it as [.., n/32, 32, ..]
    n = x.shape[dim]
    pad = (-n) % 32
    if pad:
        shp = list(x.shape); shp[dim] = pad
        x = torch.cat([x, x.new_zeros(shp)], dim=dim)
    shp = list(x.shape)
    shp[dim:dim + 1] = [shp[dim] // 32, 32]
    return x.reshape(shp), n


GUARD_SCORE = 1024.0
guard_log = []


def _blk32(x):
    """block-scaled e4m3 along the channel axis: one scale per (row, 32 channels)"""
    shp = x.shape
    return e4m3(x.reshape(*shp[:-1], shp[-1] // 32, 32), -1).reshape(shp)


def make_guarded(qk_low):
    """fp8 cross products of P V (and, qk_low, of Q' K^T) in the layers the guard lets through, split3 in the others"""
    split3 = make_attention("split3")

    def sc_attention(feat, compat, Wq, bq, Wk, bk, Wv, bv):
        C = feat.shape[-1]
        sq = 1.02 * float(torch.linalg.matrix_norm(Wq.reshape(C, C), 2)); sk = 1.02 * float(torch.linalg.matrix_norm(Wk.reshape(C, C), 2))
        F = float(feat.norm(dim=-1).max())
        guarded = (sq * F + float(bq.norm())) * (sk * F + float(bk.norm())) / math.sqrt(C) > GUARD_SCORE
        guard_log.append(int(guarded))
        if guarded:
            return split3(feat, compat, Wq, bq, Wk, bk, Wv, bv)
        q, k, v = O._lin(feat, Wq, bq), O._lin(feat, Wk, bk), O._lin(feat, Wv, bv)
        q, k, v = q.float().double(), k.float().double(), v.float().double()
        qh, kh, vh = f16(q), f16(k), f16(v)
        ql, kl, vl = f16(q - qh), f16(k - kh), f16(v - vh)
        s = qh @ kh.transpose(1, 2)
        if qk_low:
            s = s + _blk32(q) @ _blk32(kl).transpose(1, 2) + _blk32(ql) @ _blk32(k).transpose(1, 2)
        else:
            s = s + qh @ kl.transpose(1, 2) + ql @ kh.transpose(1, 2)
        s = s.float().double() / math.sqrt(C)
        z = (compat * s).float().double()
        p = torch.exp(z - z.amax(-1, keepdim=True)).float().double()
        ph = f16(p); pl = f16(p - ph)
        fix = lambda x: (x * 256.0).float().to(torch.float8_e4m3fn).double() / 256.0
        p8, pl8 = fix(p), fix(pl)
        vt, _ = tiles(v, 1); vlt, _ = tiles(vl, 1)
        v8 = e4m3(vt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
        vl8 = e4m3(vlt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
        return (ph @ vh + p8 @ vl8 + pl8 @ v8).float().double() / (ph + pl8).sum(-1, keepdim=True)
    return sc_attention


def make_attention(scheme):
    if scheme in ("fp8pv_g", "fp8x_g"):
        return make_guarded(scheme == "fp8x_g")
    lowp = {"fp8x": e4m3, "fp6x": e2m3, "fp8qk": e4m3, "fp8pv": e4m3, "fp8pv_c": e4m3, "fp6pv_c": e2m3, "fp8x_c": e4m3}.get(scheme)
    qk_low = scheme in ("fp8x", "fp6x", "fp8qk", "fp8x_c")
    pv_low = scheme in ("fp8x", "fp6x", "fp8pv")

    def sc_attention(feat, compat, Wq, bq, Wk, bk, Wv, bv):
        C = feat.shape[-1]
        q, k, v = O._lin(feat, Wq, bq), O._lin(feat, Wk, bk), O._lin(feat, Wv, bv)
        q, k, v = q.float().double(), k.float().double(), v.float().double()      # the kernel's inputs are fp32 numbers
        qh, kh, vh = f16(q), f16(k), f16(v)
        ql, kl, vl = f16(q - qh), f16(k - kh), f16(v - vh)
        s = qh @ kh.transpose(1, 2)
        if scheme == "split3" or (lowp and not qk_low):
            s = s + qh @ kl.transpose(1, 2) + ql @ kh.transpose(1, 2)
        elif lowp:
            s = s + lowp(q, -1) @ lowp(kl, -1).transpose(1, 2) + lowp(ql, -1) @ lowp(k, -1).transpose(1, 2)
        s = s.float().double() / math.sqrt(C)
        z = (compat * s).float().double()
        p = torch.exp(z - z.amax(-1, keepdim=True)).float().double()
        ph = f16(p); pl = f16(p - ph)
        if scheme == "one":
            return (ph @ vh) / ph.sum(-1, keepdim=True)
        if scheme == "p1c":          # probabilities as ONE fp16 plane, the row sum taken over the rounded values; V keeps both planes
            return (ph @ vh + ph @ vl).float().double() / ph.sum(-1, keepdim=True)
        if scheme == "fp6pv_c":      # e2m3 cross operands, every plane block-scaled (P: a query row's 32 keys of a tile), consistent row sum
            pt, n = tiles(p, 2); plt, _ = tiles(pl, 2)
            p8 = e2m3(pt, 3).reshape(p.shape[0], p.shape[1], -1)[:, :, :n]
            pl8 = e2m3(plt, 3).reshape(p.shape[0], p.shape[1], -1)[:, :, :n]
            vt, _ = tiles(v, 1); vlt, _ = tiles(vl, 1)
            v8 = e2m3(vt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
            vl8 = e2m3(vlt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
            o = ph @ vh + p8 @ vl8 + pl8 @ v8
            return o.float().double() / (ph + pl8).sum(-1, keepdim=True)
        if scheme in ("fp8pv_c", "fp8x_c"):      # P planes as UNSCALED e4m3 of (256 p) and of 256 (p - ph); row sum over ph + pl8; V block-scaled
            fix = lambda x: (x * 256.0).float().to(torch.float8_e4m3fn).double() / 256.0
            p8, pl8 = fix(p), fix(pl)
            vt, _ = tiles(v, 1); vlt, _ = tiles(vl, 1)
            v8 = e4m3(vt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
            vl8 = e4m3(vlt, 2).reshape(v.shape[0], -1, v.shape[2])[:, :v.shape[1]]
            o = ph @ vh + p8 @ vl8 + pl8 @ v8
            return o.float().double() / (ph + pl8).sum(-1, keepdim=True)
        o = ph @ vh
        if scheme == "split3" or not pv_low:
            o = o + ph @ vl + pl @ vh
        else:
            # blocks of the scaled MFMA: 32 keys of one query row (A operand) / of one feature column (B operand)
            pt, n = tiles(p, 2); plt, _ = tiles(pl, 2)
            vt, _ = tiles(v, 1); vlt, _ = tiles(vl, 1)
            p8 = lowp(pt, 3).reshape(p.shape[0], p.shape[1], -1); pl8 = lowp(plt, 3).reshape(p.shape[0], p.shape[1], -1)
            v8 = lowp(vt, 2).reshape(v.shape[0], -1, v.shape[2]); vl8 = lowp(vlt, 2).reshape(v.shape[0], -1, v.shape[2])
            o = o + (p8 @ vl8 + pl8 @ v8)
        return o.float().double() / p.sum(-1, keepdim=True)
    return sc_attention


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    kind = args[0] if len(args) > 0 else "3dmatch"
    N = int(args[1]) if len(args) > 1 else 1000
    seeds = [int(a) for a in args[2:]] or [1000, 1001, 1002, 1003]
    sigma_d = 0.1 if kind == "3dmatch" else 1.2
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)
    if "--cond" in flags:          # the KITTI-conditioned weight set (synthetic.kitti_conditioned; golden F22)
        sd = synthetic.kitti_conditioned(sd)
    schemes = ("split3", "fp8pv_c", "fp8x_c", "fp6pv_c") if "--qk" in flags else ("split3", "fp8pv_c", "fp6pv_c")
    if "--guard" in flags:
        schemes = ("split3", "fp8pv_c", "fp8x_c", "fp8pv_g", "fp8x_g")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    exact = O.sc_attention
    print(f"{kind} N={N}: max |logit - fp64 logit| per scheme")
    for seed in seeds:
        b = synthetic.synthetic_batch([seed], N=N, T=196, kind=kind)
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        compat64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], sigma_d)
        compat32, _ = O.compat_matrix(b["src_keypts"], b["tgt_keypts"], sigma_d)
        truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat64, b64["p_tokens"], b64["q_tokens"], 12))
        ref32 = O.classifier(sd, O.encoder(sd, b["corr_pos"], compat32, b["p_tokens"], b["q_tokens"], 12))
        row = [f"seed {seed}: fp32 oracle {float((ref32.double() - truth).abs().max()):.2e}"]
        try:
            for scheme in schemes:
                O.sc_attention = make_attention(scheme)
                # the compat term as the kernel sees it: the fp32 cache
                got = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat32.double(), b64["p_tokens"], b64["q_tokens"], 12))
                row.append(f"{scheme} {float((got - truth).abs().max()):.2e}" + (f" [{sum(guard_log)}]" if scheme.endswith("_g") else ""))
                guard_log.clear()
        finally:
            O.sc_attention = exact
        print("  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
