"""Weight packing: reference ``state_dict`` tensors -> the blobs the HIP kernels stream.

Every dense weight W [out, in] becomes its "P32 image" (gmf_amd/csrc/mfma_core.hpp):
``W.reshape(out/32, 32, in/8, 2, 4).permute(0, 2, 3, 1, 4)`` - 1 KiB per (32 outputs, 8 inputs)
piece in exactly the order a wavefront's lanes consume it as an MFMA operand, so a 16 KiB stage
is one contiguous LDS-DMA burst.  Eval-mode BatchNorm is folded into the preceding conv1x1
(PointDSC.py:104-109,13-21), and the softmax scales (1/sqrt(C), 1/sqrt(d_head)) times log2(e)
are folded into the Q projections so the kernels use exp2 directly.

Blob layouts (floats), one entry per kernel (see gmf_amd/csrc/encoder_kernels.hip):
  front : wst 65536 = Wp' | Wq' | Wk | Wv           vec 1664 = bp' | bq' | bk | bv | b0 | W0 image
  tail  : wst 20480 = Wa' | Wb' | Wc                vec  256 = ba' | bb' | bc
  ctx   : wst 16384 = Wk | Wv (to_kv halves)        vec  768 = content taps w0|w1|w2|b | gamma_c | beta_c
  attn  : wst 16384 = Wq'' | Wo                     vec  896 = query taps w0|w1|w2|b | gamma | beta | bo
  ff    : wst 196608 = 16 x (W1 value | W1 gate | W2 chunk)   vec 1408 = gamma | beta | b1 value | b1 gate | b2
  head  : wst 8192 = Wc1 | Wc2 (padded)             vec  128 = b1 | b2 | w3 | b3
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import _lib

LOG2E = 1.4426950408889634
C = 128
DH = 64

FRONT_WST, FRONT_VEC = 65536, 1664
TAIL_WST, TAIL_VEC = 20480, 256
CTX_WST, CTX_VEC = 16384, 768
ATTN_WST, ATTN_VEC = 16384, 896
FF_WST, FF_VEC = 196608, 1408
HEAD_WST, HEAD_VEC = 8192, 128


def p32(W: torch.Tensor) -> torch.Tensor:
    """[M, K] (M % 32 == 0, K % 8 == 0) -> flat P32 image."""
    M, K = W.shape
    assert M % 32 == 0 and K % 8 == 0, (M, K)
    return W.reshape(M // 32, 32, K // 8, 2, 4).permute(0, 2, 3, 1, 4).reshape(-1)


def _frag_index(K: int, device):
    """Feature index visited by (k-step s, K-half h, element j) of a K-wide fragment: f = 8s + j is the fragment
    index, feature = 32*(f >> 4) + 8*((f & 15) >> 2) + 4h + (f & 3)   (gmf_amd/csrc/mfma_core.hpp)."""
    s_ = torch.arange(K // 16, device=device)[:, None, None]
    h_ = torch.arange(2, device=device)[None, :, None]
    j_ = torch.arange(8, device=device)[None, None, :]
    f = 8 * s_ + j_
    return 32 * (f >> 4) + 8 * ((f & 15) >> 2) + 4 * h_ + (f & 3)          # [K/16, 2, 8]


FP16_MAX = 65504.0


def split_fp16(W: torch.Tensor, what: str):
    """W -> (hi, lo) fp16 planes with W ~ hi + lo.  Raises when a value does not fit the fp16 range: the planes of a
    weight with |w| > 65504 (|w| > 255.8 for the images stored as 256 w) would be inf / nan and so would every logit.
    The reference is plain fp32 and has no such limit; a BatchNorm with a tiny running_var can fold to weights this large
    ("Supported value range")."""
    amax = float(W.abs().max()) if W.numel() else 0.0
    if not (amax <= FP16_MAX):          # also catches nan / inf
        raise ValueError(
            f"gmf_amd.packing: {what} has max |value| = {amax:.6g} after folding/scaling, outside the fp16 range "
            f"(65504; 255.8 for weights stored as 256 w) of the split-fp16 MFMA operands.  Rescale the checkpoint "
            f"(e.g. BatchNorm running_var), or select the fp32-MFMA path with gmf_set_tuning('scattn_variant', 0).")
    hi = W.to(torch.float16)
    lo = (W - hi.float()).to(torch.float16)
    return hi, lo


def p32_h2(W: torch.Tensor) -> torch.Tensor:
    """[M, K] fp32 (M % 32 == 0, K % 16 == 0) -> flat split-fp16 image (4 B per element), as float32 words.

    Per 32-output block: [plane hi|lo][k-step][lane = (h, i)][8 fp16]; W ~ hi + lo, both round-to-nearest-even."""
    M, K = W.shape
    assert M % 32 == 0 and K % 16 == 0, (M, K)
    hi, lo = split_fp16(W, "dense weight")
    idx = _frag_index(K, W.device)
    g = torch.stack([hi, lo])[:, :, idx]                                    # [2, M, S, 2, 8]
    g = g.reshape(2, M // 32, 32, K // 16, 2, 8).permute(1, 0, 3, 4, 2, 5)  # [mb, plane, S, h, i, 8]
    return g.contiguous().view(torch.int16).reshape(-1).view(torch.float32)


H2_SCALE = 256.0     # = 1 / kH2Inv (gmf_amd/csrc/enc_common.hpp)


def p32_h2s(W: torch.Tensor) -> torch.Tensor:
    """p32_h2 of 256 W: the lo plane fp16(256 w - hi) stays a normal fp16 number for |w| >= 2^-11 instead of a subnormal
    (spacing 2^-24: an unscaled weight of 0.006 kept 16 significant bits).  The kernels fold 2^-8 into their bias add."""
    return p32_h2(W * H2_SCALE)


def conv_image(W: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """[COUT, CIN, KS, KS] fp32 (BatchNorm folded) -> split-fp16 image of 256 W for the convolution kernels (image_kernels.hip).

    k-steps of 16 input channels: k = tap * CIN + cin, or - for the stride-1 3x3 shapes with CIN == COUT, whose kernel stages
    one 16-channel block of activations at a time - k = (cin // 16) * 9 * 16 + tap * 16 + cin % 16; per 64-channel half of the output, k-step and 32-output block a (hi, lo) pair of
    1 KiB units: unit (((half * NK + ks) * 2 + blk) * 2 + plane) * 64 + lane, lane = 32 h + i holding
    W[64 half + 32 blk + i][16 ks + 8 h .. + 7]."""
    co, ci, kh, kw = W.shape
    assert co % 64 == 0 and ci % 16 == 0 and kh == kw
    Wt = (W.float() * H2_SCALE).permute(0, 2, 3, 1)                         # [co, dy, dx, cin]
    if kh == 3 and stride == 1 and co == ci:
        Wk = Wt.reshape(co, 9, ci // 16, 16).permute(0, 2, 1, 3).reshape(co, 9 * ci)      # (channel block, tap, c)
    else:
        Wk = Wt.reshape(co, kh * kw * ci)                                   # (tap, cin)
    hi, lo = split_fp16(Wk, "BatchNorm-folded convolution weight (x 256)")
    nk = Wk.shape[1] // 16
    g = torch.stack([hi, lo]).reshape(2, co // 64, 2, 32, nk, 2, 8)         # [plane, half, blk, i, ks, h, e]
    g = g.permute(1, 4, 2, 0, 5, 3, 6)                                      # [half, ks, blk, plane, h, i, e]
    return g.contiguous().view(torch.int16).reshape(-1).view(torch.float32)


def stem_image(W: torch.Tensor) -> torch.Tensor:
    """[64, 3, 7, 7] fp32 (BatchNorm folded) -> split-fp16 image of 256 W for k_stem_h2 (image_kernels.hip).

    K = 7 kernel rows x 24 (k = 24 ky + 3 kx + c; the 3 slots behind each row's 21 taps are zero) padded to 176 = 11 k-steps;
    16-byte unit ((ks * 2 + blk) * 2 + plane) * 64 + lane, lane = 32 h + j holding W[32 blk + j][16 ks + 8 h .. + 7]."""
    co, ci, kh, kw = W.shape
    assert (co, ci, kh, kw) == (64, 3, 7, 7), W.shape
    Wk = torch.zeros(64, 176, dtype=torch.float32, device=W.device)
    Wk[:, :168].view(64, 7, 24)[:, :, :21] = (W.float() * H2_SCALE).permute(0, 2, 3, 1).reshape(64, 7, 21)   # [co, ky, (kx, c)]
    hi, lo = split_fp16(Wk, "BatchNorm-folded stem weight (x 256)")
    g = torch.stack([hi, lo]).reshape(2, 2, 32, 11, 2, 8)                    # [plane, blk, j, ks, h, e]
    g = g.permute(3, 1, 0, 4, 2, 5)                                          # [ks, blk, plane, h, j, e]
    return g.contiguous().view(torch.int16).reshape(-1).view(torch.float32)


def fold_bn(W, b, sd, p, eps=1e-5):
    """conv1x1 (W [out,in], b) followed by eval BatchNorm `p` -> equivalent (W', b')."""
    scale = sd[p + "weight"] * torch.rsqrt(sd[p + "running_var"] + eps)
    return W * scale[:, None], (b - sd[p + "running_mean"]) * scale + sd[p + "bias"]


def _f(t):
    return t.detach().to(torch.float32)


def _taps(w, b):
    w = _f(w)
    return torch.cat([w[:, 0, 0], w[:, 0, 1], w[:, 0, 2], _f(b)])


def fusion_dims(sd: Dict[str, torch.Tensor], prefix: str):
    """(latent_dim, d_head) of a FusionLayer / PerceiverIO state dict; raises for widths without a HIP kernel."""
    a = prefix + "cross_attend_blocks.0."
    wq, wkv, wo = sd[a + "fn.to_q.weight"], sd[a + "fn.to_kv.weight"], sd[a + "fn.to_out.weight"]
    dh, lat = int(wq.shape[0]), int(wq.shape[1])
    ok = (lat, dh) in ((128, 64), (256, 128)) and tuple(wkv.shape) == (2 * dh, C) and tuple(wo.shape) == (lat, dh)
    if not ok:
        raise NotImplementedError(
            f"gmf_amd: HIP fusion kernels exist for (latent_dim, d_head) = (128, 64) and (256, 128) with context dim {C} "
            f"and to_out: d_head -> latent_dim; got to_q {tuple(wq.shape)}, to_kv {tuple(wkv.shape)}, to_out "
            f"{tuple(wo.shape)} (no fallback path exists)")
    return lat, dh


def pack_fusion(sd: Dict[str, torch.Tensor], prefix: str, pe: bool, img=None):
    """One FusionLayer / PerceiverIO (depth=0) -> dict of 6 blobs (fusion_layer.py:131-201, perceiver_io.py:139-221).

    Stage order matches the kernels: ctx = Wk blocks | Wv blocks; attn = Wq'' blocks (each block's K-groups in
    order, so a 256-wide input naturally spans two 16 KiB stages) | Wo blocks; ff = per 32-unit chunk
    W1 value | W1 gate | W2 column block."""
    img = img or p32          # p32 (fp32 images) or p32_h2s (split-fp16 images of 256 W): same sizes, same stage order
    img_w1 = p32_h2 if img is p32_h2s else img     # the GEGLU W1 images stay unscaled (enc_common.hpp, kH2Inv)
    lat, dh = fusion_dims(sd, prefix)
    a = prefix + "cross_attend_blocks.0."
    f = prefix + "cross_attend_blocks.1."
    dev = sd[a + "fn.to_q.weight"].device
    wkv = _f(sd[a + "fn.to_kv.weight"])
    ctx_wst = torch.cat([img(wkv[:dh]), img(wkv[dh:])])
    ctx_vec = torch.cat([_taps(sd[prefix + "cpe.proj_content.weight"], sd[prefix + "cpe.proj_content.bias"]) if pe
                         else torch.zeros(4 * C, device=dev),
                         _f(sd[a + "norm_context.weight"]), _f(sd[a + "norm_context.bias"])])
    wq = _f(sd[a + "fn.to_q.weight"]) * (dh ** -0.5 * LOG2E)
    attn_wst = torch.cat([img(wq), img(_f(sd[a + "fn.to_out.weight"]))])
    attn_vec = torch.cat([_taps(sd[prefix + "cpe.proj_q.weight"], sd[prefix + "cpe.proj_q.bias"]) if pe
                          else torch.zeros(4 * lat, device=dev),
                          _f(sd[a + "norm.weight"]), _f(sd[a + "norm.bias"]), _f(sd[a + "fn.to_out.bias"])])
    W1, b1 = _f(sd[f + "fn.net.0.weight"]), _f(sd[f + "fn.net.0.bias"])
    W2, b2 = _f(sd[f + "fn.net.2.weight"]), _f(sd[f + "fn.net.2.bias"])
    hid = W2.shape[1]
    assert tuple(W1.shape) == (2 * hid, lat) and tuple(W2.shape) == (lat, hid) and hid == 4 * lat
    chunks = []
    for c in range(hid // 32):
        chunks += [img_w1(W1[32 * c:32 * c + 32]), img_w1(W1[hid + 32 * c:hid + 32 * c + 32]), img(W2[:, 32 * c:32 * c + 32])]
    ff_wst = torch.cat(chunks)
    ff_vec = torch.cat([_f(sd[f + "norm.weight"]), _f(sd[f + "norm.bias"]), b1[:hid], b1[hid:], b2])
    out = {"ctx_wst": ctx_wst, "ctx_vec": ctx_vec, "attn_wst": attn_wst, "attn_vec": attn_vec,
           "ff_wst": ff_wst, "ff_vec": ff_vec, "latent_dim": lat, "d_head": dh}
    if (lat, dh) == (128, 64):
        assert ctx_wst.numel() == CTX_WST and ctx_vec.numel() == CTX_VEC
        assert attn_wst.numel() == ATTN_WST and attn_vec.numel() == ATTN_VEC
        assert ff_wst.numel() == FF_WST and ff_vec.numel() == FF_VEC
    else:
        assert ctx_wst.numel() == 8 * 4096 and attn_wst.numel() == 16 * 4096 and ff_wst.numel() == 192 * 4096
        assert attn_vec.numel() == 7 * 256 and ff_vec.numel() == 2816
    return {k: (v.contiguous() if torch.is_tensor(v) else v) for k, v in out.items()}


def pack_front(sd, layer: int, with_layer0: bool, identity_pointcn: bool = False, img=None):
    """PointCN_layer_i (BN folded) + projection_{q,k,v} (+ layer0) (PointDSC.py:88,104-109,23-25)."""
    n = f"encoder.blocks.NonLocal_layer_{layer}."
    pc = f"encoder.blocks.PointCN_layer_{layer}."
    dev = sd[n + "projection_q.weight"].device
    if identity_pointcn:
        Wp, bp = torch.eye(C, device=dev), torch.zeros(C, device=dev)
    else:
        Wp, bp = fold_bn(_f(sd[pc + "0.weight"])[:, :, 0], _f(sd[pc + "0.bias"]), sd, pc + "1.")
    cq = LOG2E / math.sqrt(C)
    Wq, bq = _f(sd[n + "projection_q.weight"])[:, :, 0] * cq, _f(sd[n + "projection_q.bias"]) * cq
    Wk, bk = _f(sd[n + "projection_k.weight"])[:, :, 0], _f(sd[n + "projection_k.bias"])
    Wv, bv = _f(sd[n + "projection_v.weight"])[:, :, 0], _f(sd[n + "projection_v.bias"])
    img = img or p32
    wst = torch.cat([img(Wp), img(Wq), img(Wk), img(Wv)])
    if with_layer0:
        W0 = _f(sd["encoder.layer0.weight"])[:, :, 0]
        assert W0.shape[1] <= 8, "in_dim > 8 is not supported by the layer0 MFMA prologue"
        W0p = torch.zeros(C, 8, device=dev)
        W0p[:, :W0.shape[1]] = W0
        extra = torch.cat([_f(sd["encoder.layer0.bias"]), p32(W0p)])
    else:
        extra = torch.zeros(C + 1024, device=dev)
    vec = torch.cat([bp, bq, bk, bv, extra])
    assert wst.numel() == FRONT_WST and vec.numel() == FRONT_VEC
    return wst.contiguous(), vec.contiguous()


def pack_tail(sd, layer: int, img=None):
    """fc_message with both BatchNorms folded (PointDSC.py:13-21); img = p32 (fp32 images) or p32_h2 (split-fp16)."""
    img = img or p32
    p = f"encoder.blocks.NonLocal_layer_{layer}.fc_message."
    Wa, ba = fold_bn(_f(sd[p + "0.weight"])[:, :, 0], _f(sd[p + "0.bias"]), sd, p + "1.")
    Wb, bb = fold_bn(_f(sd[p + "3.weight"])[:, :, 0], _f(sd[p + "3.bias"]), sd, p + "4.")
    Wc, bc = _f(sd[p + "6.weight"])[:, :, 0], _f(sd[p + "6.bias"])
    wst = torch.cat([img(Wa), img(Wb), img(Wc)])
    vec = torch.cat([ba, bb, bc])
    assert wst.numel() == TAIL_WST and vec.numel() == TAIL_VEC
    return wst.contiguous(), vec.contiguous()


def pack_head(sd):
    """classification head (PointDSC.py:175-181)."""
    W1, b1 = _f(sd["classification.0.weight"])[:, :, 0], _f(sd["classification.0.bias"])
    W2, b2 = _f(sd["classification.2.weight"])[:, :, 0], _f(sd["classification.2.bias"])
    w3, b3 = _f(sd["classification.4.weight"])[0, :, 0], _f(sd["classification.4.bias"])
    dev = W1.device
    wst = torch.cat([p32(W1), p32(W2), torch.zeros(HEAD_WST - 4096 - 1024, device=dev)])
    vec = torch.cat([b1, b2, w3, b3, torch.zeros(HEAD_VEC - 97, device=dev)])
    return wst.contiguous(), vec.contiguous()


def _no_object():
    return None


class _Owned:
    """Owner of one library object: never duplicated (a deep copy or a pickle of the module that holds it gets None and
    packs again on its next forward), freed exactly once."""
    _free = None

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_object, ())

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                getattr(self._lib, self._free)(self._p)
                self._p = None
        except Exception:
            pass


class PackedFusion(_Owned):
    """One FusionLayer / PerceiverIO packed by `gmf_fusion_pack_weights` (C ABI): the device pointers
    `gmf_fusion_layer_forward` takes, as plain integers (None where a split-fp16 image does not exist)."""
    _free = "gmf_packed_fusion_free"

    def __init__(self, sd: Dict[str, torch.Tensor], pe: bool, device, prefix: str = ""):
        import ctypes as C
        dev = torch.device(device)
        lib = _lib.load_library()
        self._lib = lib
        hd = _lib.handle_for(dev.index if dev.index is not None else torch.cuda.current_device()) if dev.type == "cuda" else None
        arr, keep = _lib.tensor_list(sd)
        out = C.c_void_p()
        rc = lib.gmf_fusion_pack_weights(hd.h if hd else None, arr, len(arr), prefix.encode(), 1 if pe else 0, _lib.GMF_PACK_HOST_BLOCK, C.byref(out))
        if rc != 0:
            msg = lib.gmf_last_error_string(hd.h).decode() if hd else ""
            if rc == -2:
                raise NotImplementedError(f"gmf_amd: gmf_fusion_pack_weights: {msg} (no fallback path exists)")
            raise RuntimeError(f"gmf_amd: gmf_fusion_pack_weights failed (status {rc}): {msg}")
        del keep
        self._p = out
        self._block = None
        if hd is not None:       # [r4] the block in a torch tensor (gmf_packed_fusion_place): see PackedEncoder
            nbytes = int(lib.gmf_packed_fusion_bytes(out))
            self._block = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            rc = lib.gmf_packed_fusion_place(hd.h, out, C.c_void_p(self._block.data_ptr()), nbytes, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"gmf_amd: gmf_packed_fusion_place failed (status {rc}): {lib.gmf_last_error_string(hd.h).decode()}")
        w = lib.gmf_packed_fusion_weights(out).contents
        self.latent_dim, self.d_head, self.split_fp16 = int(w.latent_dim), int(w.d_head), bool(w.split_fp16)
        for k in ("ctx_wst", "ctx_vec", "attn_wst", "attn_vec", "ff_wst", "ff_vec", "ctx_wst_h2", "attn_wst_h2", "ff_wst_h2"):
            setattr(self, k, getattr(w, k))
        if not self.split_fp16:
            import warnings
            warnings.warn(f"gmf_amd.packing: a dense weight has max |value| = {w.max_abs_scaled:.6g} after folding/scaling, outside the "
                          "fp16 range of the split-fp16 MFMA operands.  This FusionLayer runs on the fp32 MFMA.", RuntimeWarning)


class PackedEncoder(_Owned):
    """All blobs of NonLocalNet + classifier in one library-owned device block, packed by the C ABI
    (`gmf_encoder_pack_weights`, gmf_amd/csrc/gmf_pack.cpp): this class only lists the state_dict tensors by name and keeps
    the handle.  `struct` is the `gmf_encoder_weights` the forward entry points take.  (The pure-Python packers above produce
    the same blobs bit for bit - tests/test_abi_host.py - and are kept as that cross-check.)"""
    _free = "gmf_packed_encoder_free"

    def __init__(self, sd: Dict[str, torch.Tensor], num_layers: int, device, standalone_block: bool = False):
        import ctypes as C
        self.num_layers = num_layers
        dev = torch.device(device)
        on_gpu = dev.type == "cuda"
        lib = _lib.load_library()
        self._lib = lib
        hd = _lib.handle_for(dev.index if dev.index is not None else torch.cuda.current_device()) if on_gpu else None
        arr, keep = _lib.tensor_list(sd)
        out = C.c_void_p()
        # [r4] packed on the HOST (h = NULL) and placed into a torch tensor: the block lives in torch's caching allocator (visible in
        # its statistics, freed stream-ordered, no hipMalloc per re-pack and no device-wide hipFree from __del__ - ADVICE r3)
        rc = lib.gmf_encoder_pack_weights(hd.h if hd else None, arr, len(arr), int(num_layers),
                                          (_lib.GMF_PACK_STANDALONE_BLOCK if standalone_block else 0) | _lib.GMF_PACK_HOST_BLOCK, C.byref(out))
        if rc != 0:
            msg = lib.gmf_last_error_string(hd.h).decode() if hd else ""
            if rc == -2:
                raise NotImplementedError(f"gmf_amd: gmf_encoder_pack_weights: {msg} (no fallback path exists)")
            raise RuntimeError(f"gmf_amd: gmf_encoder_pack_weights failed (status {rc}): {msg}")
        del keep
        self._p = out
        self._block = None
        if on_gpu:
            nbytes = int(lib.gmf_packed_encoder_bytes(out))
            self._block = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.gmf_packed_encoder_place(hd.h, out, C.c_void_p(self._block.data_ptr()), nbytes, C.c_void_p(st))
            if rc != 0:
                raise RuntimeError(f"gmf_amd: gmf_packed_encoder_place failed (status {rc}): {lib.gmf_last_error_string(hd.h).decode()}")
        sig, sig_d, split, amax = C.c_float(), C.c_float(), C.c_int(), C.c_float()
        lib.gmf_packed_encoder_info(out, C.byref(sig), C.byref(sig_d), C.byref(split), C.byref(amax))
        self.sigma, self.sigma_d = float(sig.value), float(sig_d.value)      # read once here: no per-call host sync
        self.split_fp16 = bool(split.value)
        if not self.split_fp16:
            import warnings
            warnings.warn(f"gmf_amd.packing: a dense weight has max |value| = {amax.value:.6g} after folding/scaling, outside the fp16 "
                          "range (65504; 255.8 for weights stored as 256 w) of the split-fp16 MFMA operands.  Falling back to the "
                          "fp32-MFMA kernels for the whole encoder (about 3x slower).", RuntimeWarning)
        self.struct = lib.gmf_packed_encoder_weights(out)       # POINTER(EncoderWeights) into the packed object


PV_GUARD_SCORE = 1024.0      # gmf_pack.cpp kPvGuardScore


def pv_guard_thresholds(sd: Dict[str, torch.Tensor], num_layers: int) -> torch.Tensor:
    """Cross-check of gmf_pack.cpp `pv_guard_threshold` (gmf_encoder_weights::pv_guard): per layer the largest squared row norm
    F^2 of the layer input for which (|Wq|_2 F + |bq|)(|Wk|_2 F + |bk|) / sqrt(C) <= PV_GUARD_SCORE, with the spectral norms
    from an SVD and the packer's 2 % margin (PointDSC.py:23-25,56-64)."""
    out = []
    for i in range(num_layers):
        n = f"encoder.blocks.NonLocal_layer_{i}."
        Wq, Wk = _f(sd[n + "projection_q.weight"])[:, :, 0].double(), _f(sd[n + "projection_k.weight"])[:, :, 0].double()
        nq, nk = float(_f(sd[n + "projection_q.bias"]).double().norm()), float(_f(sd[n + "projection_k.bias"]).double().norm())
        sq, sk = 1.02 * float(torch.linalg.matrix_norm(Wq, 2)), 1.02 * float(torch.linalg.matrix_norm(Wk, 2))
        a, b, c = sq * sk, sq * nk + sk * nq, nq * nk - PV_GUARD_SCORE * math.sqrt(C)
        if c >= 0:
            out.append(-1.0)
        elif a <= 0:
            out.append(min(1e30, (c / b) ** 2) if b > 0 else 1e30)
        else:
            out.append(min(1e30, ((-b + math.sqrt(b * b - 4 * a * c)) / (2 * a)) ** 2))
    return torch.tensor(out, dtype=torch.float32)


def python_packed_encoder(sd: Dict[str, torch.Tensor], num_layers: int, standalone_block: bool = False):
    """The encoder's blobs by the pure-Python packers (host tensors): the cross-check of the C packer.  Returns
    (dict name -> tensor, split_fp16)."""
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if v.is_floating_point()}
    f1 = pack_fusion(sd, "encoder.fusion_layer_1.", pe=False) if "encoder.fusion_layer_1.cross_attend_blocks.0.fn.to_q.weight" in sd else None
    f2 = [pack_fusion(sd, f"encoder.blocks.NonLocal_layer_{i}.fusion_layer_2.", pe=True) for i in range(num_layers)]
    fronts = [pack_front(sd, i, with_layer0=(i == 0 and "encoder.layer0.weight" in sd),
                         identity_pointcn=standalone_block) for i in range(num_layers)]
    tails = [pack_tail(sd, i) for i in range(num_layers)]
    st = lambda lst: torch.stack(lst).contiguous() if lst else torch.zeros(1)
    t = {
        "ctx_wst": st([f["ctx_wst"] for f in f2]), "ctx_vec": st([f["ctx_vec"] for f in f2]),
        "attn_wst": st([f["attn_wst"] for f in f2]), "attn_vec": st([f["attn_vec"] for f in f2]),
        "ff_wst": st([f["ff_wst"] for f in f2]), "ff_vec": st([f["ff_vec"] for f in f2]),
        "front_wst": st([f[0] for f in fronts]), "front_vec": st([f[1] for f in fronts]),
        "tail_wst": st([x[0] for x in tails]), "tail_vec": st([x[1] for x in tails]),
    }
    if f1 is not None:
        for k, v in f1.items():
            if torch.is_tensor(v):
                t["f1_" + k] = v
    split = True
    try:
        h2 = {}
        if f1 is not None:
            f1h = pack_fusion(sd, "encoder.fusion_layer_1.", pe=False, img=p32_h2s)
            for k in ("ctx_wst", "attn_wst", "ff_wst"):
                h2["f1_" + k + "_h2"] = f1h[k]
        if num_layers > 0:
            f2h = [pack_fusion(sd, f"encoder.blocks.NonLocal_layer_{i}.fusion_layer_2.", pe=True, img=p32_h2s) for i in range(num_layers)]
            for k in ("ctx_wst", "attn_wst", "ff_wst"):
                h2[k + "_h2"] = torch.stack([f[k] for f in f2h]).contiguous()
            h2["front_wst_h2"] = torch.stack([pack_front(sd, i, with_layer0=(i == 0 and "encoder.layer0.weight" in sd),
                                                         identity_pointcn=standalone_block, img=p32_h2s)[0]
                                              for i in range(num_layers)]).contiguous()
            h2["tail_wst_h2"] = torch.stack([pack_tail(sd, i, img=p32_h2s)[0] for i in range(num_layers)]).contiguous()
        t.update(h2)
    except ValueError:
        split = False
    if "classification.0.weight" in sd:
        t["head_wst"], t["head_vec"] = pack_head(sd)
    return t, split
