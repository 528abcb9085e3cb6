"""CPU oracle for the GMF multimodal-fusion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the algorithm of the
reference's hot path, written from the math of each
reference function.  It exists to CHECK the HIP path: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `gmf_amd/` imports it, and the product path never falls
back to it.

Parity status: PINNED.  `oracle/gen_fixtures.py` runs the reference's own
modules (imported read-only from /root/reference in the build container) on
seeded inputs and stores their outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against those vectors.

Weights are a flat ``dict[str, Tensor]`` that uses the reference's
``state_dict`` key names unchanged (SURVEY.md section 8b "weight contract").

All file:line citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# seeded weights / inputs live in the product package (bench.py needs them without the oracle)
from gmf_amd.synthetic import (  # noqa: E402,F401
    fusion_layer_shapes, pointdsc_shapes, seeded_state_dict, random_rotation, synthetic_pair,
    synthetic_tokens, synthetic_batch)


# --------------------------------------------------------------------------
# Fusion layer pieces (fusion_layer.py)
# --------------------------------------------------------------------------
def layer_norm(x, w, b, eps: float = 1e-5):
    """nn.LayerNorm over the last dim, as used by PreNorm (fusion_layer.py:32-37,44-50)."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def conv_pos_enc_1(x, w, b):
    """Depthwise k=3 zero-padded conv along the token axis plus identity (fusion_layer.py:97-128).

    x [B,L,C]; w [C,1,3]; b [C].  out[l] = x[l] + b + w0*x[l-1] + w1*x[l] + w2*x[l+1].
    """
    xp = F.pad(x, (0, 0, 1, 1))
    w0, w1, w2 = w[:, 0, 0], w[:, 0, 1], w[:, 0, 2]
    return x + b + w0 * xp[:, :-2] + w1 * xp[:, 1:-1] + w2 * xp[:, 2:]


def cross_attention(xn, cn, Wq, Wkv, Wo, bo):
    """Single-head cross attention (fusion_layer.py:71-94, heads=1).  xn/cn are already normed."""
    d = Wq.shape[0]
    q = xn @ Wq.t()
    kv = cn @ Wkv.t()
    k, v = kv[..., :d], kv[..., d:]
    sim = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    p = torch.softmax(sim, dim=-1)
    return (p @ v) @ Wo.t() + bo


def geglu_ff(xn, W1, b1, W2, b2):
    """Linear -> GEGLU (value * gelu_erf(gate)) -> Linear (fusion_layer.py:54-69)."""
    h = xn @ W1.t() + b1
    half = h.shape[-1] // 2
    g = h[..., :half] * F.gelu(h[..., half:])
    return g @ W2.t() + b2


def fusion_layer(sd: SD, prefix: str, data, queries, pe: bool):
    """FusionLayer.forward with depth=0 (fusion_layer.py:172-201).

    data = context tokens [B,T,dim]; queries [B,N,latent_dim]; returns [B,N,latent_dim].
    """
    x = queries
    if pe:
        x = conv_pos_enc_1(x, sd[prefix + "cpe.proj_q.weight"], sd[prefix + "cpe.proj_q.bias"])
        data = conv_pos_enc_1(data, sd[prefix + "cpe.proj_content.weight"], sd[prefix + "cpe.proj_content.bias"])
    a = prefix + "cross_attend_blocks.0."
    xn = layer_norm(x, sd[a + "norm.weight"], sd[a + "norm.bias"])
    cn = layer_norm(data, sd[a + "norm_context.weight"], sd[a + "norm_context.bias"])
    x = cross_attention(xn, cn, sd[a + "fn.to_q.weight"], sd[a + "fn.to_kv.weight"],
                        sd[a + "fn.to_out.weight"], sd[a + "fn.to_out.bias"]) + x
    f = prefix + "cross_attend_blocks.1."
    xn = layer_norm(x, sd[f + "norm.weight"], sd[f + "norm.bias"])
    x = geglu_ff(xn, sd[f + "fn.net.0.weight"], sd[f + "fn.net.0.bias"],
                 sd[f + "fn.net.2.weight"], sd[f + "fn.net.2.bias"]) + x
    return x


def multihead_attention(xn, cn, Wq, Wkv, Wo, bo, heads: int):
    """Attention.forward with `heads` heads (fusion_layer.py:82-94): 'b n (h d) -> (b h) n d', softmax(q k^T d^-1/2) v, heads
    concatenated, to_out.  xn / cn are already normed."""
    B, N, _ = xn.shape
    T = cn.shape[1]
    inner = Wq.shape[0]
    d = inner // heads
    q = (xn @ Wq.t()).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kv = cn @ Wkv.t()
    k = kv[..., :inner].reshape(B, T, heads, d).permute(0, 2, 1, 3)
    v = kv[..., inner:].reshape(B, T, heads, d).permute(0, 2, 1, 3)
    p = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, inner)
    return o @ Wo.t() + bo


def fusion_layer_general(sd: SD, prefix: str, data, queries, pe: bool, depth: int, cross_heads: int, latent_heads: int):
    """FusionLayer.forward / PerceiverIO.forward for any constructor arguments (fusion_layer.py:172-201): the cross-attention block
    with `cross_heads` heads, then `depth` x {latent self-attention with `latent_heads` heads over the normed x, feed-forward}
    (test infrastructure for golden F23; GMF itself only builds depth 0 with one head)."""
    x = queries
    if pe:
        x = conv_pos_enc_1(x, sd[prefix + "cpe.proj_q.weight"], sd[prefix + "cpe.proj_q.bias"])
        data = conv_pos_enc_1(data, sd[prefix + "cpe.proj_content.weight"], sd[prefix + "cpe.proj_content.bias"])
    a = prefix + "cross_attend_blocks.0."
    xn = layer_norm(x, sd[a + "norm.weight"], sd[a + "norm.bias"])
    cn = layer_norm(data, sd[a + "norm_context.weight"], sd[a + "norm_context.bias"])
    x = multihead_attention(xn, cn, sd[a + "fn.to_q.weight"], sd[a + "fn.to_kv.weight"], sd[a + "fn.to_out.weight"],
                            sd[a + "fn.to_out.bias"], cross_heads) + x

    def ff(p, x):
        xn = layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
        return geglu_ff(xn, sd[p + "fn.net.0.weight"], sd[p + "fn.net.0.bias"], sd[p + "fn.net.2.weight"], sd[p + "fn.net.2.bias"]) + x
    x = ff(prefix + "cross_attend_blocks.1.", x)
    for i in range(depth):
        p = f"{prefix}layers.{i}.0."
        xn = layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
        x = multihead_attention(xn, xn, sd[p + "fn.to_q.weight"], sd[p + "fn.to_kv.weight"], sd[p + "fn.to_out.weight"],
                                sd[p + "fn.to_out.bias"], latent_heads) + x
        x = ff(f"{prefix}layers.{i}.1.", x)
    return x


# --------------------------------------------------------------------------
# PointDSC encoder (models/PointDSC.py:10-143)
# --------------------------------------------------------------------------
def compat_matrix(src, tgt, sigma_d):
    """Spatial-consistency matrix (PointDSC.py:216-221).  src,tgt [B,N,3] -> [B,N,N]."""
    ds = torch.cdist(src, src, compute_mode="donot_use_mm_for_euclid_dist")
    dt = torch.cdist(tgt, tgt, compute_mode="donot_use_mm_for_euclid_dist")
    d = ds - dt
    return torch.clamp(1.0 - d * d / (sigma_d * sigma_d), min=0), ds


def _lin(x, w, b):  # conv1x1 on token-major rows: x [B,N,Cin], w [Cout,Cin,1]
    return x @ w[:, :, 0].t() + b


_BN_TRAIN = [False]      # True inside training_losses(): BatchNorm1d in train() mode = statistics of the batch (all rows), biased variance


def _bn_eval(x, sd, p, eps: float = 1e-5):
    if _BN_TRAIN[0]:
        xf = x.reshape(-1, x.shape[-1])
        mu, var = xf.mean(0), xf.var(0, unbiased=False)
        return (x - mu) * torch.rsqrt(var + eps) * sd[p + "weight"] + sd[p + "bias"]
    return (x - sd[p + "running_mean"]) * torch.rsqrt(sd[p + "running_var"] + eps) * sd[p + "weight"] + sd[p + "bias"]


def sc_attention(feat, compat, Wq, bq, Wk, bk, Wv, bv):
    """Spatial-consistency self-attention (PointDSC.py:56-64), token-major.

    feat [B,N,C] -> message [B,N,C];  P = softmax_j(compat_ij * <q_i,k_j>/sqrt(C)).
    """
    C = feat.shape[-1]
    q, k, v = _lin(feat, Wq, bq), _lin(feat, Wk, bk), _lin(feat, Wv, bv)
    s = (q @ k.transpose(1, 2)) / math.sqrt(C)
    p = torch.softmax(compat * s, dim=-1)
    return p @ v


def fc_message(x, sd, p):
    """conv-BN-ReLU-conv-BN-ReLU-conv (PointDSC.py:13-21), token-major rows."""
    x = torch.relu(_bn_eval(_lin(x, sd[p + "0.weight"], sd[p + "0.bias"]), sd, p + "1."))
    x = torch.relu(_bn_eval(_lin(x, sd[p + "3.weight"], sd[p + "3.bias"]), sd, p + "4."))
    return _lin(x, sd[p + "6.weight"], sd[p + "6.bias"])


def sc_attention_heads(feat, compat, Wq, bq, Wk, bk, Wv, bv, heads: int):
    """PointDSC.py:56-64 with `heads` heads: channels [h (C / heads)] ('bhco, bhci -> bhoi' / sqrt(C / heads)), the compat matrix
    broadcast over the heads, heads concatenated (test infrastructure for golden F24; GMF builds one head)."""
    B, N, C = feat.shape
    d = C // heads
    q = _lin(feat, Wq, bq).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    k = _lin(feat, Wk, bk).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    v = _lin(feat, Wv, bv).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    p = torch.softmax(compat[:, None] * (q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, N, C)


def nonlocal_block(sd: SD, p: str, feat, compat, image_feat, heads: int = 1):
    """NonLocalBlock.forward (PointDSC.py:40-74) on token-major feat [B,N,C]."""
    qkv = (sd[p + "projection_q.weight"], sd[p + "projection_q.bias"], sd[p + "projection_k.weight"], sd[p + "projection_k.bias"],
           sd[p + "projection_v.weight"], sd[p + "projection_v.bias"])
    msg = sc_attention(feat, compat, *qkv) if heads == 1 else sc_attention_heads(feat, compat, *qkv, heads)
    msg = fc_message(msg, sd, p + "fc_message.")
    img = fusion_layer(sd, p + "fusion_layer_2.", image_feat, feat, pe=True)
    return msg + img


def point_cn(sd: SD, p: str, feat):
    """conv1x1 + BN(eval) + ReLU (PointDSC.py:104-109)."""
    return torch.relu(_bn_eval(_lin(feat, sd[p + "0.weight"], sd[p + "0.bias"]), sd, p + "1."))


def encoder(sd: SD, corr_pos, compat, p_tok, q_tok, num_layers: int):
    """NonLocalNet.forward after the image encoders (PointDSC.py:137-143).  Returns [B,N,C]."""
    image_feat = fusion_layer(sd, "encoder.fusion_layer_1.", p_tok, q_tok, pe=False)
    feat = _lin(corr_pos, sd["encoder.layer0.weight"], sd["encoder.layer0.bias"])
    for i in range(num_layers):
        feat = point_cn(sd, f"encoder.blocks.PointCN_layer_{i}.", feat)
        feat = nonlocal_block(sd, f"encoder.blocks.NonLocal_layer_{i}.", feat, compat, image_feat)
    return feat


def classifier(sd: SD, feat):
    """classification head (PointDSC.py:175-181,241) -> inlier logits [B,N]."""
    x = torch.relu(_lin(feat, sd["classification.0.weight"], sd["classification.0.bias"]))
    x = torch.relu(_lin(x, sd["classification.2.weight"], sd["classification.2.bias"]))
    return _lin(x, sd["classification.4.weight"], sd["classification.4.bias"])[..., 0]


# --------------------------------------------------------------------------
# Pose head (PointDSC.py:243-257, 268-286, 303-448, 493-528; common.py:10-75)
# --------------------------------------------------------------------------
def pick_seeds(src_dist, scores, R: float, max_num: int):
    """Parallel NMS + top-S (PointDSC.py:268-286).  B must be 1."""
    assert scores.shape[0] == 1
    ge = scores.t() >= scores                # [N,N] : score_i >= score_j
    far = src_dist[0] >= R
    is_max = (ge | far).all(dim=-1).float()
    return torch.argsort(scores * is_max, dim=1, descending=True)[:, :max_num]


def knn_indices(x, k: int):
    """k nearest (excluding rank 0) under 2 - 2 x x^T for unit rows (common.py:53-75)."""
    d = 2 - 2 * (x @ x.transpose(1, 2))
    return d.topk(k + 1, dim=-1, largest=False)[1][:, :, 1:]


def knn(x, k: int, ignore_self: bool = False, normalized: bool = True):
    """The reference's knn with all its flags (common.py:53-75): x [bs, N, C] -> [bs, N, k] indices."""
    inner = 2 * torch.matmul(x, x.transpose(2, 1))
    if normalized:
        d = 2 - inner
    else:
        xx = torch.sum(x ** 2, dim=-1, keepdim=True)
        d = xx - inner + xx.transpose(2, 1)
    if not ignore_self:
        return d.topk(k=k, dim=-1, largest=False)[1]
    return d.topk(k=k + 1, dim=-1, largest=False)[1][:, :, 1:]


def power_iteration(M, iters: int):
    """Leading eigenvector with global allclose early exit (PointDSC.py:437-448).  M [n,k,k] -> [n,k]."""
    v = torch.ones_like(M[:, :, :1])
    last = v
    for _ in range(iters):
        v = torch.bmm(M, v)
        v = v / (torch.norm(v, dim=1, keepdim=True) + 1e-6)
        if torch.allclose(v, last):
            break
        last = v
    return v[..., 0]


def rigid_transform_3d(A, Bp, w=None, weight_threshold: float = 0.0):
    """Weighted Kabsch (common.py:10-50).  A,Bp [n,k,3], w [n,k] -> T [n,4,4]."""
    if w is None:
        w = torch.ones_like(A[:, :, 0])
    w = torch.where(w < weight_threshold, torch.zeros_like(w), w)
    sw = w.sum(1, keepdim=True)[:, :, None] + 1e-6
    ca = (A * w[:, :, None]).sum(1, keepdim=True) / sw
    cb = (Bp * w[:, :, None]).sum(1, keepdim=True) / sw
    H = (A - ca).transpose(1, 2) @ (w[:, :, None] * (Bp - cb))
    U, _, Vh = torch.linalg.svd(H)
    V = Vh.transpose(1, 2)
    D = torch.eye(3, dtype=A.dtype).repeat(A.shape[0], 1, 1)
    D[:, 2, 2] = torch.det(V @ U.transpose(1, 2))
    R = V @ D @ U.transpose(1, 2)
    t = cb.transpose(1, 2) - R @ ca.transpose(1, 2)
    T = torch.eye(4, dtype=A.dtype).repeat(A.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3:4] = t
    return T


def seed_weights(feat_n, src, tgt, knn_idx, sigma: float, sigma_d: float, iters: int):
    """Per-seed second-order compatibility + leading eigenvector weights (PointDSC.py:335-365).

    feat_n [B,N,C] unit rows; knn_idx [B,S,k] -> weights [B*S,k], src_knn, tgt_knn [B*S,k,3]
    """
    B, S, k = knn_idx.shape
    bi = torch.arange(B)[:, None, None]
    f = feat_n[bi, knn_idx]                         # [B,S,k,C]
    Mf = torch.clamp(1 - (1 - f @ f.transpose(2, 3)) / sigma ** 2, min=0)
    sk, tk = src[bi, knn_idx], tgt[bi, knn_idx]     # [B,S,k,3]
    d = torch.cdist(sk, sk, compute_mode="donot_use_mm_for_euclid_dist") - \
        torch.cdist(tk, tk, compute_mode="donot_use_mm_for_euclid_dist")
    Ms = torch.clamp(1 - d * d / sigma_d ** 2, min=0)
    M = (Mf * Ms).reshape(B * S, k, k).clone()
    idx = torch.arange(k)
    M[:, idx, idx] = 0
    v = power_iteration(M, iters).reshape(B, S, k)
    w = v / (v.sum(-1, keepdim=True) + 1e-6)
    return w.reshape(B * S, k), sk.reshape(B * S, k, 3), tk.reshape(B * S, k, 3)


def score_hypotheses(T, src, tgt, tau: float):
    """Inlier-ratio fitness of S hypotheses + argmax (PointDSC.py:413-425).  T [B,S,4,4]."""
    pred = torch.einsum("bsnm,bkm->bskn", T[:, :, :3, :3], src) + T[:, :, None, :3, 3]
    dist = torch.norm(pred - tgt[:, None], dim=-1)          # [B,S,N]
    fit = (dist < tau).float().mean(-1)
    best = fit.argmax(dim=1)
    bi = torch.arange(T.shape[0])
    return fit, T[bi, best], (dist[bi, best] < tau).float()


def cal_seed_trans(feat_n, src, tgt, seeds, sigma: float, sigma_d: float, k: int, iters: int, tau: float):
    """PointDSC.cal_seed_trans (PointDSC.py:303-427)."""
    B, N, _ = feat_n.shape
    k = min(k, N - 1)
    knn_all = knn_indices(feat_n, k)
    knn_idx = torch.gather(knn_all, 1, seeds[:, :, None].expand(-1, -1, k))
    w, sk, tk = seed_weights(feat_n, src, tgt, knn_idx, sigma, sigma_d, iters)
    Ts = rigid_transform_3d(sk, tk, w).reshape(B, -1, 4, 4)
    fit, final_T, labels = score_hypotheses(Ts, src, tgt, tau)
    return Ts, fit, final_T, labels, knn_idx


def post_refinement(T, src, tgt, tau: float, max_iters: int = 20):
    """IRLS refinement with early exit on unchanged inlier count (PointDSC.py:493-528).  B must be 1.

    The reference hard-wires the threshold list to 0.10 when inlier_threshold == 0.10
    and 1.2 otherwise (PointDSC.py:505-508).
    """
    assert T.shape[0] == 1
    thr = 0.10 if tau == 0.10 else 1.2
    prev = 0
    for _ in range(max_iters):
        warped = src @ T[:, :3, :3].transpose(1, 2) + T[:, None, :3, 3]
        d = torch.norm(warped - tgt, dim=-1)
        inl = (d < thr)[0]
        n = int(inl.sum())
        if abs(n - prev) < 1:
            break
        prev = n
        T = rigid_transform_3d(src[:, inl], tgt[:, inl], (1 / (1 + (d / thr) ** 2))[:, inl])
    return T


def pointdsc_forward(sd: SD, data: dict, num_layers: int = 12, ratio: float = 0.1, k: int = 40,
                     num_iterations: int = 10, inlier_threshold: float = 0.10, nms_radius: float = 0.10,
                     testing: bool = True):
    """PointDSC.forward (PointDSC.py:191-266) with image tokens fed directly (ResNet is upstream).

    data: corr_pos [B,N,6], src_keypts, tgt_keypts [B,N,3], p_tokens, q_tokens [B,T,C].
    Test mode loops pairs one at a time as the reference requires (PointDSC.py:279,504).
    Returns dict with logits [B,N], final_trans [B,4,4], final_labels [B,N], corr_features.
    """
    src, tgt = data["src_keypts"], data["tgt_keypts"]
    sigma_d = float(sd["sigma_spat"])
    sigma = float(sd["sigma"])
    compat, src_dist = compat_matrix(src, tgt, sigma_d)
    feat = encoder(sd, data["corr_pos"], compat, data["p_tokens"], data["q_tokens"], num_layers)
    feat_n = F.normalize(feat, p=2, dim=-1)
    logits = classifier(sd, feat)
    B, N = logits.shape
    S = int(N * ratio)
    finals, labels, seeds_all = [], [], []
    if testing:
        for b in range(B):
            sl = slice(b, b + 1)
            seeds = pick_seeds(src_dist[sl], logits[sl], nms_radius, S)
            _, _, fT, lab, _ = cal_seed_trans(feat_n[sl], src[sl], tgt[sl], seeds, sigma, sigma_d, k,
                                              num_iterations, inlier_threshold)
            fT = post_refinement(fT, src[sl], tgt[sl], inlier_threshold)
            finals.append(fT), labels.append(lab), seeds_all.append(seeds)
        final_T, final_labels, seeds = torch.cat(finals), torch.cat(labels), torch.cat(seeds_all)
    else:
        seeds = torch.argsort(logits, dim=1, descending=True)[:, :S]
        _, _, final_T, _, _ = cal_seed_trans(feat_n, src, tgt, seeds, sigma, sigma_d, k, num_iterations,
                                             inlier_threshold)
        final_labels = logits
    return {"logits": logits, "final_trans": final_T, "final_labels": final_labels,
            "corr_features": feat, "seeds": seeds}


# --------------------------------------------------------------------------
# DGR surface (GMF_DeepGlobalRegistration/*/core/registration.py:91-113)
# --------------------------------------------------------------------------
def weighted_procrustes(X, Y, w, eps: float):
    """DGR weighted Procrustes with an fp64 3x3 SVD.  X,Y [N,3], w [N,1] -> R [3,3], t [3] (fp32)."""
    W1 = w.abs().sum()
    wn = w / (W1 + eps)
    mux = (wn * X).sum(0, keepdim=True)
    muy = (wn * Y).sum(0, keepdim=True)
    Sxy = ((Y - muy).t() @ (wn * (X - mux))).double()
    U, _, Vh = torch.linalg.svd(Sxy)
    D = torch.eye(3, dtype=torch.float64)
    if torch.det(U) * torch.det(Vh) < 0:
        D[2, 2] = -1
    R = (U @ D @ Vh).float()
    t = (muy[0] - (R @ mux.t())[:, 0]).float()
    return R, t


def dgr_inlier_weights(logits, clip: float = 0.05):
    """sigmoid then zero weights below `clip` (core/deep_global_registration.py:323-325)."""
    w = torch.sigmoid(logits)
    return torch.where(w < clip, torch.zeros_like(w), w)


# --------------------------------------------------------------------------
# DGR robust pose refinement:
# GlobalRegistration (core/registration.py:135-194), Transformation (:116-132),
# ortho2rotation (:16-63), HighDimSmoothL1Loss (core/loss.py:42-61)
# --------------------------------------------------------------------------
def ortho2rotation(poses):
    """6-D rotation parameters [B,6] -> rotation matrices [B,3,3] (Gram-Schmidt; registration.py:16-63)."""
    x_raw, y_raw = poses[:, 0:3], poses[:, 3:6]
    x = x_raw / torch.clamp(torch.sqrt((x_raw ** 2).sum(1, keepdim=True)), min=1e-8)
    factor = (x * y_raw).sum(1, keepdim=True) / torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8)
    v = y_raw - factor * x
    y = v / torch.clamp(torch.sqrt((v ** 2).sum(1, keepdim=True)), min=1e-8)
    z = torch.stack((x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1],
                     x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2],
                     x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]), 1)
    return torch.stack((x, y, z), 2)


def high_dim_smooth_l1(X, Y, weights, w1, quantization_size: float, eps: float):
    """loss.py:51-61: per-point 0.5*d^2 for d^2 < 1, else 0.5*(sqrt(d^2 + eps) - 0.5); weighted mean."""
    sq_dist = torch.sum(((X - Y) / quantization_size) ** 2, dim=1, keepdim=True)
    use_sq_half = 0.5 * (sq_dist < 1).float()
    loss = (0.5 - use_sq_half) * (torch.sqrt(sq_dist + eps) - 0.5) + use_sq_half * sq_dist
    if weights is None:
        return loss.mean()
    return (loss * weights).sum() / w1


def global_registration(points, trans_points, weights=None, max_iter: int = 1000, max_break_count: int = 20,
                        break_threshold_ratio: float = 1e-5, quantization_size: float = 1.0):
    """Adam (lr 0.1, exponential decay 0.999) on a 6-D rotation + translation initialised by the (weighted) Procrustes
    solution, stopped when the relative loss change stayed below `break_threshold_ratio` `max_break_count` times
    (the counter never resets, registration.py:181-184) or the loss drops below 1e-7.
    Returns R [3,3], t [3], {"iterations", "loss", "break_count"} like the reference."""
    eps = float(torch.finfo(torch.float32).eps)
    if weights is None:
        R, t = weighted_procrustes(points, trans_points, torch.ones(points.shape[0], 1), 0.0)   # = argmin_se3_squared_dist
        w1 = None
    else:
        weights = weights.detach()
        R, t = weighted_procrustes(points, trans_points, weights, eps)
        w1 = weights.sum()
    rot6d = torch.cat((R[:, 0], R[:, 1]))[None].clone().requires_grad_(True)
    trans = t[None].clone().requires_grad_(True)

    def forward():
        return points @ ortho2rotation(rot6d)[0].t() + trans

    opt = torch.optim.Adam([rot6d, trans], lr=1e-1)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999)
    with torch.no_grad():
        loss_prev = high_dim_smooth_l1(forward(), trans_points, weights, w1, quantization_size, eps).item()
    break_counter, i, loss = 0, 0, None
    for i in range(max_iter):
        loss = high_dim_smooth_l1(forward(), trans_points, weights, w1, quantization_size, eps)
        if loss.item() < 1e-7:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if abs(loss_prev - loss.item()) < loss_prev * break_threshold_ratio:
            break_counter += 1
            if break_counter >= max_break_count:
                break
        loss_prev = loss.item()
    return (ortho2rotation(rot6d.detach())[0], trans.detach()[0],
            {"iterations": i, "loss": loss.item(), "break_count": break_counter})


# --------------------------------------------------------------------------
# Descriptor matching
# --------------------------------------------------------------------------
def nn_match_pointdsc(src_desc, tgt_desc):
    """GMF_PointDSC/datasets/ThreeDMatch.py:164-166 / demo_registration.py:101-103 for unit descriptors."""
    distance = torch.sqrt(2 - 2 * (src_desc @ tgt_desc.t()) + 1e-6)
    dis, idx = distance.min(dim=1)
    return idx, dis


def find_knn_dgr(F0, F1, nn_max_n: int = -1):
    """DGR find_knn_gpu with knn=1 (core/knn.py:23-74; pdist core/metrics.py:62-69): chunks of nn_max_n rows use the
    L2 distance sqrt(D2 + 1e-7), the unchunked path the squared distance."""
    def d2(a, b):
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(2)
    if nn_max_n > 1:
        ds, ix = [], []
        for i in range(0, len(F0), nn_max_n):
            dist = torch.sqrt(d2(F0[i:i + nn_max_n], F1) + 1e-7)
            m, j = dist.min(dim=1, keepdim=True)
            ds.append(m), ix.append(j)
        return torch.cat(ix), torch.cat(ds)
    m, j = d2(F0, F1).min(dim=1)
    return j, m[:, None]


# --------------------------------------------------------------------------
# Validation step (forward half): the feature-similarity matrix the non-test forward
# returns and the three metrics libs/trainer.py:194-262 evaluates on it.
# --------------------------------------------------------------------------
def similarity_matrix(feat_n, sigma: float):
    """PointDSC.py:231-234: M = clamp(1 - (1 - <f_i, f_j>) / sigma^2, 0, 1) with a zero diagonal."""
    M = torch.matmul(feat_n, feat_n.permute(0, 2, 1))
    M = torch.clamp(1 - (1 - M) / sigma ** 2, min=0, max=1)
    idx = torch.arange(M.shape[1])
    M[:, idx, idx] = 0
    return M


def classification_loss(pred, gt, balanced: bool = True, weight=None):
    """libs/loss.py:67-113 (ClassificationLoss).  BCE-with-logits over the whole batch, `pos_weight` = #neg / #pos with
    both counts floored at one; precision / recall / f1 are those of pair 0 only (loss.py:99-101, sklearn's binary
    scores, 0 when undefined); the mean logits are over the whole batch.  Returns a dict of floats."""
    gt = gt.float()
    num_pos = torch.relu(gt.sum() - 1) + 1
    num_neg = torch.relu((1 - gt).sum() - 1) + 1
    sp_pos = F.softplus(-pred)        # -log sigmoid(x)
    sp_neg = F.softplus(pred)         # -log (1 - sigmoid(x))
    if weight is not None:
        loss = ((gt * sp_pos + (1 - gt) * sp_neg) * weight).mean()
    elif not balanced:
        loss = (gt * sp_pos + (1 - gt) * sp_neg).mean()
    else:
        loss = ((num_neg / num_pos) * gt * sp_pos + (1 - gt) * sp_neg).mean()
    g0, p0 = gt[0] > 0.5, pred[0] > 0
    tp, fp, fn = float((g0 & p0).sum()), float((~g0 & p0).sum()), float((g0 & ~p0).sum())
    precision = tp / (tp + fp) if tp + fp > 0 else 0.0
    recall = tp / (tp + fn) if tp + fn > 0 else 0.0
    f1 = 2 * tp / (2 * tp + fp + fn) if 2 * tp + fp + fn > 0 else 0.0
    return {"loss": float(loss), "precision": precision, "recall": recall, "f1": f1,
            "logit_true": float((pred * gt).sum() / max(1.0, float(gt.sum()))),
            "logit_false": float((pred * (1 - gt)).sum() / max(1.0, float((1 - gt).sum())))}


def spectral_matching_loss(M, gt_labels, balanced: bool = True):
    """libs/loss.py:116-140 (SpectralMatchingLoss): gt_M = outer AND of the labels with a zero diagonal; balanced form
    = mean over pairs of half the mean squared miss on the inlier-inlier entries plus half that on the rest."""
    gt = gt_labels.float()
    gt_M = ((gt[:, None, :] + gt[:, :, None]) == 2).float()
    idx = torch.arange(gt_M.shape[1])
    gt_M[:, idx, idx] = 0
    if balanced:
        lp = ((M - 1) ** 2 * gt_M).sum((-1, -2)) / (torch.relu(gt_M.sum((-1, -2)) - 1.0) + 1.0)
        ln = (M ** 2 * (1 - gt_M)).sum((-1, -2)) / (torch.relu((1 - gt_M).sum((-1, -2)) - 1.0) + 1.0)
        return float(torch.mean(lp * 0.5 + ln * 0.5))
    return float(((M - gt_M) ** 2).mean())


def sm_loss_from_features(corr_features, sigma, gt_labels, balanced: bool = True):
    """Differentiable restatement (tensors in, 0-dim tensor out) of PointDSC.py:229-234 followed by libs/loss.py:116-140:
    F.normalize -> M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1) with the diagonal set to zero in place -> the loss.
    torch autograd over it is the CPU checker of the HIP backward slice (golden F17 pins it to the reference's gradients)."""
    fn = F.normalize(corr_features, p=2, dim=-1)
    M = torch.matmul(fn, fn.permute(0, 2, 1))
    M = torch.clamp(1 - (1 - M) / sigma ** 2, min=0, max=1)
    idx = torch.arange(M.shape[1])
    M[:, idx, idx] = 0
    gt = gt_labels.float()
    gt_M = ((gt[:, None, :] + gt[:, :, None]) == 2).float()
    gt_M[:, idx, idx] = 0
    if balanced:
        lp = ((M - 1) ** 2 * gt_M).sum((-1, -2)) / (torch.relu(gt_M.sum((-1, -2)) - 1.0) + 1.0)
        ln = (M ** 2 * (1 - gt_M)).sum((-1, -2)) / (torch.relu((1 - gt_M).sum((-1, -2)) - 1.0) + 1.0)
        return torch.mean(lp * 0.5 + ln * 0.5)
    return ((M - gt_M) ** 2).mean()


def training_losses(sd: SD, data: dict, num_layers: int, balanced: bool):
    """Differentiable restatement of the reference's default training objective (libs/trainer.py:131-141 with
    config_3DMatch.py:49-52): the model in train() mode (BatchNorm batch statistics) and its non-test forward
    (PointDSC.py:216-241), ClassificationLoss (loss.py:85-93) + SpectralMatchingLoss (loss.py:116-140).  Tensors in `sd` may
    require grad; returns (logits, M, class_loss, sm_loss) as tensors.  Golden F19 pins it to the reference's gradients."""
    src, tgt = data["src_keypts"], data["tgt_keypts"]
    with torch.no_grad():
        compat, _ = compat_matrix(src, tgt, float(sd["sigma_spat"]))
    _BN_TRAIN[0] = True
    try:
        feat = encoder(sd, data["corr_pos"], compat, data["p_tokens"], data["q_tokens"], num_layers)
        logits = classifier(sd, feat)
    finally:
        _BN_TRAIN[0] = False
    fn = F.normalize(feat, p=2, dim=-1)
    M = torch.matmul(fn, fn.permute(0, 2, 1))
    M = torch.clamp(1 - (1 - M) / sd["sigma"] ** 2, min=0, max=1)
    idx = torch.arange(M.shape[1])
    M[:, idx, idx] = 0
    gt = data["gt_labels"].float()
    num_pos = torch.relu(gt.sum() - 1) + 1
    num_neg = torch.relu((1 - gt).sum() - 1) + 1
    if balanced:
        cl = F.binary_cross_entropy_with_logits(logits, gt, pos_weight=num_neg / num_pos)
    else:
        cl = F.binary_cross_entropy_with_logits(logits, gt)
    gt_M = ((gt[:, None, :] + gt[:, :, None]) == 2).float()
    gt_M[:, idx, idx] = 0
    if balanced:
        lp = ((M - 1) ** 2 * gt_M).sum((-1, -2)) / (torch.relu(gt_M.sum((-1, -2)) - 1.0) + 1.0)
        ln = (M ** 2 * (1 - gt_M)).sum((-1, -2)) / (torch.relu((1 - gt_M).sum((-1, -2)) - 1.0) + 1.0)
        sm = torch.mean(lp * 0.5 + ln * 0.5)
    else:
        sm = ((M - gt_M) ** 2).mean()
    return logits, M, cl, sm


def transformation_loss(trans, gt_trans, src_keypts, tgt_keypts, probs, re_thre: float = 15.0, te_thre: float = 30.0):
    """libs/loss.py:12-64 (TransformationLoss).  As the reference, pair i's warped source points are compared with the
    target points of EVERY pair of the batch (`warp_src_keypts - tgt_keypts` broadcasts [N,3] against [bs,N,3],
    loss.py:47-48,61): rmse_i and loss_i are means over bs x N points.  Returns (loss, recall %, RE deg, TE cm, RMSE)."""
    bs = trans.shape[0]
    recall, RE, TE, RMSE, loss = 0, 0.0, 0.0, 0.0, 0.0
    for i in range(bs):
        R, t, gR, gt_t = trans[i, :3, :3], trans[i, :3, 3], gt_trans[i, :3, :3], gt_trans[i, :3, 3]
        re = torch.acos(torch.clamp((torch.trace(R.T @ gR) - 1) / 2.0, min=-1, max=1)) * 180 / math.pi
        te = torch.sqrt(((t - gt_t) ** 2).sum()) * 100
        warp = src_keypts[i] @ R.T + t
        diff = warp[None] - tgt_keypts
        if te < te_thre and re < re_thre:
            recall += 1
        RE, TE, RMSE = RE + float(re), TE + float(te), RMSE + float(diff.norm(dim=-1).mean())
        if int((probs[i] > 0).sum()) >= 1:
            loss += float((diff ** 2).sum(-1).mean())
    return loss / bs, recall * 100.0 / bs, RE / bs, TE / bs, RMSE / bs


def pose_loss_from_features(corr_features, sigma, logits, src, tgt, sigma_d: float, ratio: float = 0.1, k: int = 40,
                            iters: int = 10, tau: float = 0.10):
    """Differentiable restatement (tensors in, 0-dim tensor out) of the non-test forward's pose head followed by the loss term
    of TransformationLoss: F.normalize (PointDSC.py:229) -> top-S seeds by confidence (:246) -> cal_seed_trans (:303-427) ->
    final_trans -> (1/bs) sum_i [any(probs_i > 0)] mean |warp_i - tgt|^2 with the reference's broadcast over the batch
    (libs/loss.py:57-62).  torch autograd over it is the CPU checker of gmf_pose_head_backward +
    gmf_transformation_loss_backward (golden F20 pins it to the reference's gradients).  Returns (loss, final_trans)."""
    fn = F.normalize(corr_features, p=2, dim=-1)
    bs, N, _ = fn.shape
    seeds = torch.argsort(logits, dim=1, descending=True)[:, :int(N * ratio)]
    _, _, final_T, _, _ = cal_seed_trans(fn, src, tgt, seeds, sigma, sigma_d, k, iters, tau)
    loss = torch.zeros((), dtype=fn.dtype)
    for i in range(bs):
        if int((logits[i] > 0).sum()) >= 1:
            warp = src[i] @ final_T[i, :3, :3].T + final_T[i, :3, 3]
            loss = loss + ((warp[None] - tgt) ** 2).sum(-1).mean()
    return loss / bs, final_T
