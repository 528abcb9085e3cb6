"""FusionLayer drop-in (reference: GMF_PointDSC/models/fusion_layer.py:131-201).

Same constructor arguments, ``forward(data, mask=None, queries_encoder=None)`` signature and
``state_dict`` keys as the reference class.  The sub-modules below only HOLD parameters under the
reference's names; the eval-mode forward pass is three HIP kernels (context prepare, cross-attention,
GEGLU feed-forward) behind ``gmf_fusion_layer_forward``; in train() mode with autograd enabled the forward
is the differentiable composition of HIP training primitives in ``gmf_amd/train.py``.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib, packing
from ._util import handle_and_stream, params_version, require_cuda_f32


class _GEGLU(nn.Module):            # fusion_layer.py:54-57 (parameter-free; index 1 of FeedForward.net)
    pass


class _FeedForward(nn.Module):      # fusion_layer.py:59-69
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), _GEGLU(), nn.Linear(dim * mult, dim))


class _Attention(nn.Module):        # fusion_layer.py:71-80
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, out_to_query=False):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        # PointDSC: Linear(inner, context_dim) (fusion_layer.py:80); DGR: Linear(inner, query_dim) (perceiver_io.py:83)
        self.to_out = nn.Linear(inner, query_dim if out_to_query else context_dim)


class _PreNorm(nn.Module):          # fusion_layer.py:32-37
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None


class _ConvPosEnc(nn.Module):       # fusion_layer.py:97-116
    def __init__(self, dim_q, dim_content, k=3):
        super().__init__()
        self.proj_q = nn.Conv1d(dim_q, dim_q, k, 1, k // 2, groups=dim_q)
        self.proj_content = nn.Conv1d(dim_content, dim_content, k, 1, k // 2, groups=dim_content)


class FusionLayer(nn.Module):
    _OUT_TO_QUERY = False

    def __init__(self, depth, dim, latent_dim=512, cross_heads=1, latent_heads=8, cross_dim_head=64,
                 latent_dim_head=64, weight_tie_layers=False, pe=False):
        super().__init__()
        if depth != 0:
            raise NotImplementedError("gmf_amd.FusionLayer: latent self-attention layers (depth > 0) are never "
                                      "instantiated by GMF (PointDSC.py:31,94; resunet_new.py:518,620) and have no HIP kernel")
        if cross_heads != 1:
            raise NotImplementedError("gmf_amd.FusionLayer: GMF uses cross_heads = 1 (PointDSC.py:33,96)")
        self.pe = pe
        self.dim, self.latent_dim, self.cross_dim_head = dim, latent_dim, cross_dim_head
        if pe:
            self.cpe = _ConvPosEnc(dim_q=latent_dim, dim_content=dim)
        self.cross_attend_blocks = nn.ModuleList([
            _PreNorm(latent_dim, _Attention(latent_dim, dim, heads=cross_heads, dim_head=cross_dim_head,
                                            out_to_query=self._OUT_TO_QUERY), context_dim=dim),
            _PreNorm(latent_dim, _FeedForward(latent_dim)),
        ])
        self.layers = nn.ModuleList([])
        self._packed = None
        self._packed_version = None
        self.split_fp16_ff = True      # feed-forward on the f16 MFMA with split-fp16 operands (False: fp32 MFMA, for A/B runs)
        self.split_fp16_attn = True    # context preparation + cross-attention likewise

    def _blobs(self, device):
        """The layer's packed weights (`gmf_fusion_pack_weights`, C ABI) from the module's own state_dict, cached until a
        parameter changes."""
        ver = (params_version(self), str(device))
        if self._packed is None or self._packed_version != ver:
            self._packed = packing.PackedFusion(self.state_dict(), self.pe, device)
            self._packed_version = ver
        return self._packed

    def forward(self, data, mask=None, queries_encoder=None):
        """data [B,T,dim] context tokens; queries_encoder [B,N,latent_dim] (any strides) -> [B,N,latent_dim].

        `mask` is accepted and ignored, exactly as in the reference (fusion_layer.py:82)."""
        if self.training and torch.is_grad_enabled():
            # train() mode with autograd on: the differentiable path (gmf_amd/train.py - HIP training primitives behind a
            # torch.autograd.Function; gradients to data, queries and every parameter).  eval() keeps the fused inference kernels.
            from .train import fusion_layer_train
            return fusion_layer_train(self, data, queries_encoder)
        x = require_cuda_f32(queries_encoder, "queries_encoder")
        data = require_cuda_f32(data, "data").contiguous()
        B, N, Cq = x.shape
        T = data.shape[1]
        if data.shape[2] != packing.C:
            raise NotImplementedError(f"gmf_amd.FusionLayer: HIP kernels are built for 128-wide context tokens, got {data.shape[2]}")
        blobs = self._blobs(x.device)
        if Cq != blobs.latent_dim:
            raise RuntimeError(f"gmf_amd.FusionLayer: queries are {Cq} wide but the layer was built for {blobs.latent_dim}")
        out = torch.empty((B, N, Cq), device=x.device, dtype=torch.float32)
        h, st = handle_and_stream(x, check=True)            # (raises if an earlier forward on this device produced NaN / inf)
        h.call("gmf_fusion_layer_forward", 1 if self.pe else 0, blobs.latent_dim, blobs.d_head,
               blobs.ctx_wst, blobs.ctx_vec, blobs.attn_wst, blobs.attn_vec, blobs.ff_wst, blobs.ff_vec,
               data.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), x.stride(2),
               out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), B, N, T, st,
               blobs.ff_wst_h2 if (self.split_fp16_ff and blobs.ff_wst_h2) else None,
               *((blobs.ctx_wst_h2, blobs.attn_wst_h2) if (self.split_fp16_attn and blobs.attn_wst_h2) else (None, None)))
        return out
