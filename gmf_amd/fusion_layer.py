"""FusionLayer drop-in (reference: GMF_PointDSC/models/fusion_layer.py:131-201).

Same constructor arguments, ``forward(data, mask=None, queries_encoder=None)`` signature and
``state_dict`` keys as the reference class.  The sub-modules below only HOLD parameters under the
reference's names; the eval-mode forward pass is three HIP kernels (context prepare, cross-attention,
GEGLU feed-forward) behind ``gmf_fusion_layer_forward``; in train() mode with autograd enabled the forward
is the differentiable composition of HIP training primitives in ``gmf_amd/train.py``.

[r5] Configurations GMF itself never instantiates - latent self-attention layers (``depth`` > 0, ``weight_tie_layers``),
several cross-attention heads, widths other than the two the fused kernels are built for - run the same HIP primitives the
training path uses (``gmf_gemm_f32``, ``gmf_layernorm_forward``, ``gmf_softmax_rows``, ``gmf_geglu``, ``gmf_lcpe``) as a
forward-only composition (``FusionLayer._forward_general``): every constructor argument of the reference class is honoured.
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
        self.pe = pe
        self.dim, self.latent_dim, self.cross_dim_head = dim, latent_dim, cross_dim_head
        self.depth, self.cross_heads, self.latent_heads, self.latent_dim_head = depth, cross_heads, latent_heads, latent_dim_head
        if pe:
            self.cpe = _ConvPosEnc(dim_q=latent_dim, dim_content=dim)
        self.cross_attend_blocks = nn.ModuleList([
            _PreNorm(latent_dim, _Attention(latent_dim, dim, heads=cross_heads, dim_head=cross_dim_head,
                                            out_to_query=self._OUT_TO_QUERY), context_dim=dim),
            _PreNorm(latent_dim, _FeedForward(latent_dim)),
        ])
        # fusion_layer.py:155-170: `depth` x {latent self-attention, feed-forward}; with weight_tie_layers every layer is the SAME pair of
        # modules (cache_fn), so the state_dict lists the shared tensors under every layer's name - as the reference's does
        self.layers = nn.ModuleList([])
        tied = None
        for _ in range(depth):
            if weight_tie_layers and tied is not None:
                pair = tied
            else:
                pair = (_PreNorm(latent_dim, _Attention(latent_dim, None, heads=latent_heads, dim_head=latent_dim_head,
                                                        out_to_query=self._OUT_TO_QUERY)),
                        _PreNorm(latent_dim, _FeedForward(latent_dim)))
                tied = pair
            self.layers.append(nn.ModuleList(list(pair)))
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
        if self.training:
            raise RuntimeError("gmf_amd.FusionLayer: train() mode without autograd - call eval() for inference")
        if not self._fused_form(data.shape[2]):
            return self._forward_general(data, x)
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

    # -- [r5] every other configuration of the reference's constructor -----------------------------------------------------------
    def _fused_form(self, ctx_dim: int) -> bool:
        """The three fused kernels cover what GMF instantiates: depth 0, one cross-attention head, 128-wide context tokens and
        (latent, head) = (128, 64) [PointDSC.py:29-38,92-100] or (256, 128) [resunet_new.py:516-525]."""
        return (self.depth == 0 and self.cross_heads == 1 and ctx_dim == packing.C and self.dim == packing.C
                and (self.latent_dim, self.cross_dim_head) in ((128, 64), (256, 128)))

    def _attend(self, xn, cn, att, Bq, Nq, Tk):
        """Multi-head attention of fusion_layer.py:82-94 on normed rows: xn [Bq*Nq, Dq], cn [Bq*Tk, Dc] -> [Bq*Nq, heads*d]."""
        from . import train as P
        hds, d = att.heads, att.dim_head
        inner = hds * d
        q = P.gemm(xn, att.to_q.weight, tb=True)                    # [B N, inner]
        kv = P.gemm(cn, att.to_kv.weight, tb=True)                  # [B T, 2 inner] = k | v
        S = torch.empty((Bq, hds, Nq, Tk), device=xn.device, dtype=torch.float32)
        a = torch.empty((Bq * Nq, inner), device=xn.device, dtype=torch.float32)
        for hd in range(hds):                                       # one batched product per head (batch = B)
            P.gemm(q, kv, tb=True, out=S, m=Nq, n=Tk, k=d, lda=inner, ldb=2 * inner, ldc=Tk, a_off=hd * d, b_off=hd * d,
                   c_off=hd * Nq * Tk, batch=Bq, sa=Nq * inner, sb=Tk * 2 * inner, sc=hds * Nq * Tk)
        Pm = P.softmax_rows(S.reshape(Bq * hds * Nq, Tk), d ** -0.5)
        for hd in range(hds):
            P.gemm(Pm, kv, out=a, m=Nq, n=d, k=Tk, lda=Tk, ldb=2 * inner, ldc=inner, a_off=hd * Nq * Tk, b_off=inner + hd * d,
                   c_off=hd * d, batch=Bq, sa=hds * Nq * Tk, sb=Tk * 2 * inner, sc=Nq * inner)
        return a

    def _forward_general(self, data, x):
        """fusion_layer.py:172-201 for ANY constructor arguments, forward only, from the library's HIP primitives (no fused kernel
        exists for these shapes; GMF never builds them).  Same arithmetic as the reference: fp32 products (`gmf_gemm_f32`), exact
        erf GELU, LayerNorm eps 1e-5."""
        from . import train as P
        with torch.no_grad():
            B, N, lat = x.shape
            T, dim = data.shape[1], data.shape[2]
            if lat != self.latent_dim or dim != self.dim:
                raise RuntimeError(f"gmf_amd.FusionLayer: built for queries {self.latent_dim} / context {self.dim} wide, got {lat} / {dim}")
            x2, c2 = x.contiguous().reshape(B * N, lat), data.reshape(B * T, dim)
            if self.pe:
                x2 = P.lcpe_fwd(x2, self.cpe.proj_q.weight.contiguous(), self.cpe.proj_q.bias, N)
                c2 = P.lcpe_fwd(c2, self.cpe.proj_content.weight.contiguous(), self.cpe.proj_content.bias, T)
            ca, cf = self.cross_attend_blocks
            xn, _, _ = P.layernorm_fwd(x2, ca.norm.weight, ca.norm.bias)
            cn, _, _ = P.layernorm_fwd(c2, ca.norm_context.weight, ca.norm_context.bias)
            a = self._attend(xn, cn, ca.fn, B, N, T)
            if ca.fn.to_out.weight.shape[0] != lat:
                raise RuntimeError("gmf_amd.FusionLayer: to_out maps to the context width (fusion_layer.py:80), which the residual "
                                   f"only accepts when dim == latent_dim (got {dim} / {lat}) - as in the reference")
            x2 = P.gemm(a, ca.fn.to_out.weight, tb=True, bias=ca.fn.to_out.bias, residual=x2)

            def ff(blk, xin):
                xn2, _, _ = P.layernorm_fwd(xin, blk.norm.weight, blk.norm.bias)
                hdn = P.gemm(xn2, blk.fn.net[0].weight, tb=True, bias=blk.fn.net[0].bias)
                return P.gemm(P.geglu_fwd(hdn), blk.fn.net[2].weight, tb=True, bias=blk.fn.net[2].bias, residual=xin)
            x2 = ff(cf, x2)
            for sa, sf in self.layers:
                xn, _, _ = P.layernorm_fwd(x2, sa.norm.weight, sa.norm.bias)
                a = self._attend(xn, xn, sa.fn, B, N, N)            # self-attention: the context is the normed x (PreNorm, :44-52)
                x2 = P.gemm(a, sa.fn.to_out.weight, tb=True, bias=sa.fn.to_out.bias, residual=x2)
                x2 = ff(sf, x2)
            return x2.reshape(B, N, lat)
