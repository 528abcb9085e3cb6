"""Seeded synthetic inputs and weights (SURVEY.md section 8d "synthetic inputs").

NumPy ``default_rng`` streams only, so the same seeds give the same pairs, tokens and weights
in the fixture generator, the tests, the oracle and ``bench.py``.  Weight dicts use the
reference's ``state_dict`` key names.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Tuple

import numpy as np
import torch

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# seeded weights / inputs (shared by the fixture generator, tests and bench)
# --------------------------------------------------------------------------
def _rng_for(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def fusion_layer_shapes(prefix: str, dim: int, latent_dim: int, dim_head: int, pe: bool,
                        out_to_query: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Parameter shapes of one FusionLayer / PerceiverIO with depth=0.

    GMF_PointDSC/models/fusion_layer.py:131-170; DGR twin model/perceiver_io.py:139-180.
    `out_to_query` selects the DGR variant whose to_out maps inner -> query_dim
    (perceiver_io.py:83) instead of inner -> context_dim (fusion_layer.py:80).
    """
    inner = dim_head  # cross_heads = 1
    out_dim = latent_dim if out_to_query else dim
    ff_hidden = latent_dim * 4
    s = {}
    if pe:
        s[prefix + "cpe.proj_q.weight"] = (latent_dim, 1, 3)
        s[prefix + "cpe.proj_q.bias"] = (latent_dim,)
        s[prefix + "cpe.proj_content.weight"] = (dim, 1, 3)
        s[prefix + "cpe.proj_content.bias"] = (dim,)
    a = prefix + "cross_attend_blocks.0."
    s[a + "fn.to_q.weight"] = (inner, latent_dim)
    s[a + "fn.to_kv.weight"] = (2 * inner, dim)
    s[a + "fn.to_out.weight"] = (out_dim, inner)
    s[a + "fn.to_out.bias"] = (out_dim,)
    s[a + "norm.weight"] = (latent_dim,)
    s[a + "norm.bias"] = (latent_dim,)
    s[a + "norm_context.weight"] = (dim,)
    s[a + "norm_context.bias"] = (dim,)
    f = prefix + "cross_attend_blocks.1."
    s[f + "fn.net.0.weight"] = (2 * ff_hidden, latent_dim)
    s[f + "fn.net.0.bias"] = (2 * ff_hidden,)
    s[f + "fn.net.2.weight"] = (latent_dim, ff_hidden)
    s[f + "fn.net.2.bias"] = (latent_dim,)
    s[f + "norm.weight"] = (latent_dim,)
    s[f + "norm.bias"] = (latent_dim,)
    return s


def _bn_shapes(prefix: str, c: int) -> Dict[str, Tuple[int, ...]]:
    return {prefix + "weight": (c,), prefix + "bias": (c,),
            prefix + "running_mean": (c,), prefix + "running_var": (c,),
            prefix + "num_batches_tracked": ()}


def pointdsc_shapes(in_dim: int = 6, num_layers: int = 12, C: int = 128) -> Dict[str, Tuple[int, ...]]:
    """Non-image state_dict entries of reference PointDSC (models/PointDSC.py:146-181, 77-112, 10-38)."""
    s: Dict[str, Tuple[int, ...]] = {"sigma": (1,), "sigma_spat": (1,)}
    s["encoder.layer0.weight"] = (C, in_dim, 1)
    s["encoder.layer0.bias"] = (C,)
    s.update(fusion_layer_shapes("encoder.fusion_layer_1.", C, C, C // 2, pe=False))
    for i in range(num_layers):
        p = f"encoder.blocks.PointCN_layer_{i}."
        s[p + "0.weight"] = (C, C, 1)
        s[p + "0.bias"] = (C,)
        s.update(_bn_shapes(p + "1.", C))
        n = f"encoder.blocks.NonLocal_layer_{i}."
        s[n + "fc_message.0.weight"] = (C // 2, C, 1)
        s[n + "fc_message.0.bias"] = (C // 2,)
        s.update(_bn_shapes(n + "fc_message.1.", C // 2))
        s[n + "fc_message.3.weight"] = (C // 2, C // 2, 1)
        s[n + "fc_message.3.bias"] = (C // 2,)
        s.update(_bn_shapes(n + "fc_message.4.", C // 2))
        s[n + "fc_message.6.weight"] = (C, C // 2, 1)
        s[n + "fc_message.6.bias"] = (C,)
        for nm in ("projection_q", "projection_k", "projection_v"):
            s[n + nm + ".weight"] = (C, C, 1)
            s[n + nm + ".bias"] = (C,)
        s.update(fusion_layer_shapes(n + "fusion_layer_2.", C, C, C // 2, pe=True))
    s["classification.0.weight"] = (32, C, 1)
    s["classification.0.bias"] = (32,)
    s["classification.2.weight"] = (32, 32, 1)
    s["classification.2.bias"] = (32,)
    s["classification.4.weight"] = (1, 32, 1)
    s["classification.4.bias"] = (1,)
    return s


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, gain: float = 0.8,
                      sigma: float = 1.0, sigma_d: float = 0.1) -> SD:
    """Deterministic non-trivial weights from NumPy default_rng (stable across versions).

    Every tensor draws from its own generator keyed by (seed, crc32(name)) so the
    result does not depend on dict order.  Scales are chosen so that activations
    stay O(1) through 12 residual layers while logits span a few units
    (SURVEY.md section 7 "Default-init weights give almost constant logits").

    `gain` sets the conditioning of the whole network: the 12 blocks have no
    normalisation between them, so at gain >= 0.9 every block amplifies rounding
    differences by ~1.4x and two fp32 evaluations that merely sum in a different
    order already differ by 3e-4 on the logits (measured: fp32 vs fp64 of this
    file); at gain 0.8 that fp32 noise floor is ~2e-5, which makes the 1e-4 parity
    gate meaningful.  Tests use 0.8 for the 1e-4 gate and 0.9 for a stress case whose
    tolerance is stated as a multiple of the measured noise floor.
    """
    out: SD = {}
    for k, shp in shapes.items():
        r = _rng_for(seed, k)
        leaf = k.rsplit(".", 1)[-1]
        if k == "sigma":
            v = np.full(shp, sigma, np.float32)
        elif k == "sigma_spat":
            v = np.full(shp, sigma_d, np.float32)
        elif leaf == "num_batches_tracked":
            out[k] = torch.tensor(7, dtype=torch.int64)
            continue
        elif leaf == "running_mean":
            v = r.normal(0.0, 0.2, shp)
        elif leaf == "running_var":
            v = r.uniform(0.6, 1.6, shp)
        elif leaf == "bias":
            v = r.normal(0.0, 0.1, shp)
        elif leaf == "weight" and len(shp) == 1:  # LayerNorm / BatchNorm scale
            v = r.uniform(0.7, 1.3, shp)
        elif leaf == "weight" and ".cpe." in k:   # depthwise k=3 taps
            v = r.normal(0.0, 0.3, shp)
        else:                                      # dense weight [out, in, (1)]
            fan_in = int(np.prod(shp[1:]))         # [out, in, (k...)]: conv1x1, Linear and Conv2d alike
            g = gain
            if ".projection_q." in k or ".projection_k." in k:
                g *= 2.5                           # sharper spatial-consistency attention
            elif ".fn.to_q." in k or ".fn.to_kv." in k:
                g *= 1.6                           # sharper cross attention
            elif k.startswith("classification."):
                g *= 2.2                           # logits spanning several units
            v = r.normal(0.0, g / math.sqrt(fan_in), shp)
        out[k] = torch.from_numpy(np.asarray(v, np.float32).reshape(shp).copy())
    return out


F23_CASES = {
    # golden F23 (oracle/gen_fixtures.py): class, depth, dim, latent_dim, cross_heads, latent_heads, cross_dim_head, latent_dim_head,
    # weight_tie_layers, pe, B, N, T - configurations of FusionLayer / PerceiverIO that GMF never instantiates
    "fl_d2_h2": ("fl", 2, 128, 128, 2, 4, 32, 32, False, True, 2, 70, 33),
    "fl_tied": ("fl", 3, 128, 128, 1, 2, 64, 48, True, True, 1, 97, 40),
    "fl_w96": ("fl", 1, 96, 96, 3, 2, 16, 24, False, False, 2, 45, 12),
    "pio_d1": ("pio", 1, 128, 256, 2, 8, 64, 32, False, True, 1, 130, 50),
}


def f23_state_dict(shapes: Dict[str, Tuple[int, ...]], tied: bool) -> SD:
    """The weights of a golden-F23 module from its own key -> shape map: every tensor seeded by its key (seed 123); with
    weight_tie_layers the tensors of layers 1 .. share those of layer 0 (the reference's modules are the same objects there)."""
    import re
    sd: SD = {}
    for k, shp in shapes.items():
        src = re.sub(r"^layers\.\d+\.", "layers.0.", k) if tied else k
        sd[k] = seeded_state_dict({src: tuple(shp)}, seed=123)[src]
    return sd


def f23_inputs(name: str):
    """(queries [B,N,latent], context [B,T,dim]) of golden F23 case `name`."""
    cls, depth, dim, lat, ch, lh, cdh, ldh, tie, pe, B, N, T = F23_CASES[name]
    r = np.random.default_rng([123, N, T])
    x = torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32))
    ctx = torch.from_numpy(r.normal(0, 1, (B, T, dim)).astype(np.float32))
    return x, ctx


F24_CASES = {"c64_h2": (64, 2, 2, 90, 20), "c128_h4": (128, 4, 1, 150, 33), "c32_h1": (32, 1, 2, 40, 7)}     # golden F24: channels, heads, B, N, T


def f24_inputs(name: str):
    """(feat [B,C,N], src, tgt [B,N,3], image_feat [B,T,C]) of golden F24 case `name` (oracle/gen_fixtures.py gen_f24 draws the same stream)."""
    C, H, B, N, T = F24_CASES[name]
    r = np.random.default_rng([124, C, N])
    feat = torch.from_numpy(r.normal(0, 1, (B, C, N)).astype(np.float32))
    img = torch.from_numpy(r.normal(0, 1, (B, T, C)).astype(np.float32))
    src = torch.from_numpy(r.uniform(0, 3, (B, N, 3)).astype(np.float32))
    tgt = src + torch.from_numpy(r.normal(0, 0.05, (B, N, 3)).astype(np.float32))
    return feat, src, tgt, img


def kitti_conditioned(sd: SD, div: float = 13.0) -> SD:
    """The seeded weights re-conditioned for KITTI-shape inputs (coordinates of +-40 m instead of 0..3 m).

    `corr_pos` enters the network through `encoder.layer0` alone (PointDSC.py:139); the seeded first-layer weights are
    scaled for 3DMatch-size coordinates, so on +-40 m inputs every activation behind them is ~13x larger and the
    reference's own fp32 evaluation is 3e-4 from the exact network.  Dividing that one weight by `div` brings the
    activations back to the 3DMatch scale (fp32 floor 1.7e-5 at N = 700 / 2000, logit spread unchanged), which is what
    makes the literal 1e-4 gate testable on the sigma_d = 1.2 path (golden F22).  Returns a new dict; `sd` is not modified."""
    out = dict(sd)
    out["encoder.layer0.weight"] = sd["encoder.layer0.weight"] / div
    return out


def random_rotation(r: np.random.Generator) -> np.ndarray:
    q, _ = np.linalg.qr(r.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 2] = -q[:, 2]
    return q


def synthetic_pair(seed: int, N: int, kind: str = "3dmatch", inlier_ratio: Optional[float] = None):
    """One synthetic scene pair (SURVEY.md section 8d "synthetic inputs").

    Returns dict of float32 numpy arrays: corr_pos [N,6], src_keypts [N,3], tgt_keypts [N,3],
    gt_trans [4,4], gt_labels [N].
    """
    r = np.random.default_rng([seed, 0x5eed])
    if kind == "3dmatch":
        lo, hi = np.zeros(3), np.full(3, 3.0)
        tmag, noise = 0.5, 0.01
        ratio = 0.25 if inlier_ratio is None else inlier_ratio
    elif kind == "kitti":
        lo, hi = np.array([-40.0, -40.0, -2.0]), np.array([40.0, 40.0, 2.0])
        tmag, noise = 5.0, 0.05
        ratio = 0.40 if inlier_ratio is None else inlier_ratio
    else:
        raise ValueError(kind)
    src = r.uniform(lo, hi, (N, 3))
    R = random_rotation(r)
    t = r.uniform(-tmag, tmag, 3)
    n_in = int(round(N * ratio))
    labels = np.zeros(N, np.float32)
    inl = r.permutation(N)[:n_in]
    labels[inl] = 1
    tgt = r.uniform(lo, hi, (N, 3)) @ R.T + t  # outliers: anywhere in the (moved) box
    tgt[inl] = src[inl] @ R.T + t + r.normal(0, noise, (n_in, 3))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    corr = np.concatenate([src, tgt], 1)
    corr = corr - corr.mean(0, keepdims=True)  # datasets/ThreeDMatch.py:207-210
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return {"corr_pos": f32(corr), "src_keypts": f32(src), "tgt_keypts": f32(tgt),
            "gt_trans": f32(T), "gt_labels": labels}


def synthetic_tokens(seed: int, T: int, C: int = 128):
    r = np.random.default_rng([seed, 0x70c])
    return (r.normal(0, 1, (T, C)).astype(np.float32), r.normal(0, 1, (T, C)).astype(np.float32))


def synthetic_batch(seeds, N: int, T: int, kind: str = "3dmatch", C: int = 128):
    pairs = [synthetic_pair(s, N, kind) for s in seeds]
    toks = [synthetic_tokens(s, T, C) for s in seeds]
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in pairs]))
    return {"corr_pos": st("corr_pos"), "src_keypts": st("src_keypts"), "tgt_keypts": st("tgt_keypts"),
            "gt_trans": st("gt_trans"), "gt_labels": st("gt_labels"),
            "p_tokens": torch.from_numpy(np.stack([t[0] for t in toks])),
            "q_tokens": torch.from_numpy(np.stack([t[1] for t in toks]))}




def dgr_scene(N: int, seed: int, inlier_ratio: float = 0.3, noise: float = 0.01, clip: float = 0.05):
    """One DGR-surface problem: X ~ U[0,3)^3, Y = R X + t + noise for the inliers and uniform
    for the outliers, w = sigmoid(logit) with weights below `clip` zeroed (deep_global_registration.py:322-325).
    Returns X [N,3], Y [N,3], w [N,1] (float32 torch tensors) and the ground truth R [3,3], t [3] (numpy)."""
    import numpy as np
    import torch
    rng = np.random.default_rng([113, N, seed])
    X = rng.uniform(0, 3, (N, 3)).astype(np.float32)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    t = rng.uniform(-0.5, 0.5, 3)
    Y = (X @ Q.T + t + rng.normal(0, noise, (N, 3))).astype(np.float32)
    out = rng.random(N) > inlier_ratio
    Y[out] = rng.uniform(0, 3, (int(out.sum()), 3)).astype(np.float32)
    logit = np.where(out, rng.normal(-2, 1.5, N), rng.normal(2, 1.5, N)).astype(np.float32)
    w = (1.0 / (1.0 + np.exp(-logit.astype(np.float64)))).astype(np.float32)
    w[w < clip] = 0
    return torch.from_numpy(X), torch.from_numpy(Y), torch.from_numpy(w)[:, None], Q.astype(np.float32), t.astype(np.float32)


def seeded_images(n_images: int, H: int, W: int) -> torch.Tensor:
    """[n,3,H,W] uniform [0,1) images, the inputs of golden F15 (oracle/gen_fixtures.py gen_f15 draws the same stream)."""
    r = np.random.default_rng([115, n_images, H, W])
    return torch.from_numpy(r.uniform(0, 1, (n_images, 3, H, W)).astype(np.float32))
