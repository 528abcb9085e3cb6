"""Training path, second backward slice: forward + backward of one FusionLayer /
PerceiverIO (reference: GMF_PointDSC/models/fusion_layer.py:32-128,172-201; DGR twin model/perceiver_io.py) as a
``torch.autograd.Function`` over the training primitives of the C ABI (``gmf_gemm_f32``, ``gmf_lcpe``,
``gmf_layernorm_*``, ``gmf_softmax_rows``, ``gmf_geglu``, ``gmf_colsum`` - gmf_amd/csrc/train_kernels.hip).

Every tensor stays on the device, every kernel is HIP; torch only owns the memory and the autograd graph.  The
function is NOT the fused inference kernel: it saves its activations (x', LN statistics, q, k|v, P, the 1024-wide
hidden layer ...) for the backward, as the reference's autograd does.  Gradients equal the reference's own
(golden F18).
"""
from __future__ import annotations

import torch

from ._util import handle_and_stream, require_cuda_f32


def _p(t, off=0):
    return t.data_ptr() + 4 * off


def gemm(a, b, ta=False, tb=False, bias=None, residual=None, alpha=1.0, out=None, m=None, n=None, k=None, lda=None, ldb=None,
         ldc=None, a_off=0, b_off=0, c_off=0, batch=1, sa=0, sb=0, sc=0):
    """out = alpha * op(a) op(b) (+ bias) (+ residual) through gmf_gemm_f32.  With the keyword geometry left out, `a` and `b`
    are dense 2-D row-major matrices; with it, any sub-matrix / batch of a larger buffer (offsets and strides in floats)."""
    if m is None:
        m, k = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        n = b.shape[0] if tb else b.shape[1]
        lda, ldb = a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((batch, m, n) if batch > 1 else (m, n), device=a.device, dtype=torch.float32)
        ldc, sc = n, m * n
    h, st = handle_and_stream(a)
    h.call("gmf_gemm_f32", 1 if ta else 0, 1 if tb else 0, _p(a, a_off), _p(b, b_off), _p(out, c_off),
           None if bias is None else bias.data_ptr(), None if residual is None else _p(residual, c_off), m, n, k, lda, ldb,
           ldc, sa, sb, sc, batch, float(alpha), st)
    return out


def colsum(x, y=None, mean=None, rstd=None, shift=0, L=None):
    """[rows, C] -> [C]: sum_r x[r] * y'[r + shift] (see gmf_colsum)."""
    rows, C = x.shape
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    h, st = handle_and_stream(x)
    h.call("gmf_colsum", x.data_ptr(), None if y is None else y.data_ptr(), None if mean is None else mean.data_ptr(),
           None if rstd is None else rstd.data_ptr(), int(shift), int(L if L is not None else rows), rows, C, out.data_ptr(), st)
    return out


def layernorm_fwd(x, gamma, beta):
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    h, st = handle_and_stream(x)
    h.call("gmf_layernorm_forward", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, C, st)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_add=None):
    rows, C = x.shape
    dx = torch.empty_like(x)
    h, st = handle_and_stream(x)
    h.call("gmf_layernorm_backward", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
           None if dx_add is None else dx_add.data_ptr(), dx.data_ptr(), rows, C, st)
    return dx


def lcpe_fwd(x2d, w, b, L):
    y = torch.empty_like(x2d)
    h, st = handle_and_stream(x2d)
    h.call("gmf_lcpe", 0, x2d.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), x2d.shape[0], L, x2d.shape[1], st)
    return y


def lcpe_bwd(dy2d, w, L):
    dx = torch.empty_like(dy2d)
    h, st = handle_and_stream(dy2d)
    h.call("gmf_lcpe", 1, dy2d.data_ptr(), w.data_ptr(), None, dx.data_ptr(), dy2d.shape[0], L, dy2d.shape[1], st)
    return dx


def softmax_rows(S2d, scale):
    P = torch.empty_like(S2d)
    h, st = handle_and_stream(S2d)
    h.call("gmf_softmax_rows", 0, S2d.data_ptr(), None, P.data_ptr(), S2d.shape[0], S2d.shape[1], float(scale), st)
    return P


def softmax_rows_bwd(P2d, dP2d, scale):
    dS = torch.empty_like(P2d)
    h, st = handle_and_stream(P2d)
    h.call("gmf_softmax_rows", 1, P2d.data_ptr(), dP2d.data_ptr(), dS.data_ptr(), P2d.shape[0], P2d.shape[1], float(scale), st)
    return dS


def geglu_fwd(hdn):
    rows, H2 = hdn.shape
    g = torch.empty((rows, H2 // 2), device=hdn.device, dtype=torch.float32)
    h, st = handle_and_stream(hdn)
    h.call("gmf_geglu", 0, hdn.data_ptr(), None, g.data_ptr(), rows, H2 // 2, st)
    return g


def geglu_bwd(hdn, dg):
    dh = torch.empty_like(hdn)
    h, st = handle_and_stream(hdn)
    h.call("gmf_geglu", 1, hdn.data_ptr(), dg.data_ptr(), dh.data_ptr(), hdn.shape[0], hdn.shape[1] // 2, st)
    return dh


class _FusionLayerTrain(torch.autograd.Function):
    """forward(data [B,T,dim], x [B,N,lat], pe, *params) -> [B,N,lat] with params in the order of `_param_list`."""

    @staticmethod
    def forward(ctx, data, x, pe, *params):
        data = require_cuda_f32(data, "data").contiguous()
        x = require_cuda_f32(x, "queries_encoder").contiguous()
        params = [p.detach().contiguous() for p in params]
        if pe:
            wq_t, bq_t, wc_t, bc_t = params[:4]
            rest = params[4:]
        else:
            rest = params
        g1, b1n, gc, bcn, Wq, Wkv, Wo, bo, g2, b2n, W1, bf1, W2, bf2 = rest
        B, N, lat = x.shape
        T, dim = data.shape[1], data.shape[2]
        dh = Wq.shape[0]
        scale = dh ** -0.5
        x2, c2 = x.reshape(B * N, lat), data.reshape(B * T, dim)
        xp = lcpe_fwd(x2, wq_t, bq_t, N) if pe else x2
        cp = lcpe_fwd(c2, wc_t, bc_t, T) if pe else c2
        xn, mu1, rs1 = layernorm_fwd(xp, g1, b1n)
        cn, muc, rsc = layernorm_fwd(cp, gc, bcn)
        q = gemm(xn, Wq, tb=True)                                   # [BN, dh]
        kv = gemm(cn, Wkv, tb=True)                                 # [BT, 2 dh] = k | v
        S = torch.empty((B, N, T), device=x.device, dtype=torch.float32)
        gemm(q, kv, tb=True, out=S, m=N, n=T, k=dh, lda=dh, ldb=2 * dh, ldc=T, batch=B, sa=N * dh, sb=T * 2 * dh, sc=N * T)
        P = softmax_rows(S.reshape(B * N, T), scale)
        a = torch.empty((B * N, dh), device=x.device, dtype=torch.float32)
        gemm(P, kv, out=a, m=N, n=dh, k=T, lda=T, ldb=2 * dh, ldc=dh, b_off=dh, batch=B, sa=N * T, sb=T * 2 * dh, sc=N * dh)
        x1 = gemm(a, Wo, tb=True, bias=bo, residual=xp)             # [BN, lat]
        xn2, mu2, rs2 = layernorm_fwd(x1, g2, b2n)
        hdn = gemm(xn2, W1, tb=True, bias=bf1)                      # [BN, 2 H]
        g = geglu_fwd(hdn)
        out = gemm(g, W2, tb=True, bias=bf2, residual=x1)
        ctx.pe, ctx.dims, ctx.scale = pe, (B, N, T, lat, dim, dh), scale
        ctx.save_for_backward(x2, c2, xp, cp, mu1, rs1, muc, rsc, xn, cn, q, kv, P, a, x1, mu2, rs2, xn2, hdn, g, *params)
        return out.reshape(B, N, lat)

    @staticmethod
    def backward(ctx, dout):
        (x2, c2, xp, cp, mu1, rs1, muc, rsc, xn, cn, q, kv, P, a, x1, mu2, rs2, xn2, hdn, g, *params) = ctx.saved_tensors
        pe, scale = ctx.pe, ctx.scale
        B, N, T, lat, dim, dh = ctx.dims
        if pe:
            wq_t, bq_t, wc_t, bc_t = params[:4]
            rest = params[4:]
        else:
            rest = params
        g1, b1n, gc, bcn, Wq, Wkv, Wo, bo, g2, b2n, W1, bf1, W2, bf2 = rest
        dev = dout.device
        d2 = require_cuda_f32(dout, "grad").contiguous().reshape(B * N, lat)
        # ---- feed-forward (fusion_layer.py:54-69,191) ----
        dW2 = gemm(d2, g, ta=True)                                  # [lat, H]
        db2 = colsum(d2)
        dg = gemm(d2, W2)                                           # [BN, H]
        dhd = geglu_bwd(hdn, dg)                                    # [BN, 2 H]
        dW1 = gemm(dhd, xn2, ta=True)                               # [2 H, lat]
        db1 = colsum(dhd)
        dxn2 = gemm(dhd, W1)                                        # [BN, lat]
        dg2 = colsum(dxn2, y=x1, mean=mu2, rstd=rs2)
        dbn2 = colsum(dxn2)
        dx1 = layernorm_bwd(dxn2, x1, g2, mu2, rs2, dx_add=d2)
        # ---- cross-attention (fusion_layer.py:71-94,190) ----
        dWo = gemm(dx1, a, ta=True)                                 # [lat, dh]
        dbo = colsum(dx1)
        da = gemm(dx1, Wo)                                          # [BN, dh]
        dP = torch.empty((B * N, T), device=dev, dtype=torch.float32)
        gemm(da, kv, tb=True, out=dP, m=N, n=T, k=dh, lda=dh, ldb=2 * dh, ldc=T, b_off=dh, batch=B, sa=N * dh, sb=T * 2 * dh, sc=N * T)
        dkv = torch.empty((B * T, 2 * dh), device=dev, dtype=torch.float32)
        gemm(P, da, ta=True, out=dkv, m=T, n=dh, k=N, lda=T, ldb=dh, ldc=2 * dh, c_off=dh, batch=B, sa=N * T, sb=N * dh, sc=T * 2 * dh)   # dv
        dS = softmax_rows_bwd(P, dP, scale)
        dq = torch.empty((B * N, dh), device=dev, dtype=torch.float32)
        gemm(dS, kv, out=dq, m=N, n=dh, k=T, lda=T, ldb=2 * dh, ldc=dh, batch=B, sa=N * T, sb=T * 2 * dh, sc=N * dh)
        gemm(dS, q, ta=True, out=dkv, m=T, n=dh, k=N, lda=T, ldb=dh, ldc=2 * dh, batch=B, sa=N * T, sb=N * dh, sc=T * 2 * dh)            # dk
        dWq = gemm(dq, xn, ta=True)                                 # [dh, lat]
        dxn = gemm(dq, Wq)                                          # [BN, lat]
        dWkv = gemm(dkv, cn, ta=True)                               # [2 dh, dim]
        dcn = gemm(dkv, Wkv)                                        # [BT, dim]
        dg1 = colsum(dxn, y=xp, mean=mu1, rstd=rs1)
        dbn1 = colsum(dxn)
        dxp = layernorm_bwd(dxn, xp, g1, mu1, rs1, dx_add=dx1)
        dgc = colsum(dcn, y=cp, mean=muc, rstd=rsc)
        dbnc = colsum(dcn)
        dcp = layernorm_bwd(dcn, cp, gc, muc, rsc)
        grads = []
        if pe:
            # ---- LCPE (fusion_layer.py:118-128) ----
            dx = lcpe_bwd(dxp, wq_t, N)
            dc = lcpe_bwd(dcp, wc_t, T)
            dwq = torch.stack([colsum(dxp, y=x2, shift=s, L=N) for s in (-1, 0, 1)], dim=1).reshape(wq_t.shape)
            dwc = torch.stack([colsum(dcp, y=c2, shift=s, L=T) for s in (-1, 0, 1)], dim=1).reshape(wc_t.shape)
            grads += [dwq, colsum(dxp), dwc, colsum(dcp)]
        else:
            dx, dc = dxp, dcp
        grads += [dg1, dbn1, dgc, dbnc, dWq, dWkv, dWo, dbo, dg2, dbn2, dW1, db1, dW2, db2]
        return (dc.reshape(B, T, dim), dx.reshape(B, N, lat), None, *grads)


def _param_list(layer):
    attn, ff = layer.cross_attend_blocks
    ps = []
    if layer.pe:
        ps += [layer.cpe.proj_q.weight, layer.cpe.proj_q.bias, layer.cpe.proj_content.weight, layer.cpe.proj_content.bias]
    ps += [attn.norm.weight, attn.norm.bias, attn.norm_context.weight, attn.norm_context.bias,
           attn.fn.to_q.weight, attn.fn.to_kv.weight, attn.fn.to_out.weight, attn.fn.to_out.bias,
           ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[2].weight, ff.fn.net[2].bias]
    return ps


def fusion_layer_train(layer, data, queries):
    """Differentiable FusionLayer / PerceiverIO forward (depth = 0, one cross head): gradients flow to `data`, `queries`
    and every parameter of `layer`."""
    return _FusionLayerTrain.apply(data, queries, bool(layer.pe), *_param_list(layer))
