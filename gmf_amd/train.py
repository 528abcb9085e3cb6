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


SIGMA_ON_DEVICE = object()       # [r5] marker in place of sigma's host value: the kernels read the parameter's device memory


class sigma_device:
    """`with sigma_device(handle, sigma_tensor):` - the C calls inside read PointDSC's learnable sigma (PointDSC.py:164) from the
    tensor's device memory (gmf_set_sigma_device) instead of a by-value float; None = the by-value form.  Host-side state only:
    inside a graph capture the calls between enter and exit are what gets captured."""

    def __init__(self, h, sigma):
        self.h, self.sigma = h, sigma

    def __enter__(self):
        if self.sigma is not None:
            self.h.call("gmf_set_sigma_device", self.sigma.data_ptr())

    def __exit__(self, *exc):
        if self.sigma is not None:
            self.h.call("gmf_set_sigma_device", None)
        return False


def gemm(a, b, ta=False, tb=False, bias=None, residual=None, alpha=1.0, out=None, m=None, n=None, k=None, lda=None, ldb=None,
         ldc=None, a_off=0, b_off=0, c_off=0, batch=1, sa=0, sb=0, sc=0, relu=False):
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
           ldc, sa, sb, sc, batch, float(alpha), 1 if relu else 0, st)
    return out


def colsum(x, y=None, mean=None, rstd=None, shift=0, L=None, dual=False, relu_y=None):
    """[rows, C] -> [C]: sum_r x[r] * y'[r + shift] (see gmf_colsum).  dual: ([C] that sum, [C] the plain column sums of x)
    from ONE pass; relu_y: x is masked by the saved output of a ReLU first."""
    rows, C = x.shape
    out = torch.empty(2 * C if dual else C, device=x.device, dtype=torch.float32)
    h, st = handle_and_stream(x)
    h.call("gmf_colsum", x.data_ptr(), None if y is None else y.data_ptr(), None if mean is None else mean.data_ptr(),
           None if rstd is None else rstd.data_ptr(), None, None, 0, int(shift), int(L if L is not None else rows), rows, C,
           None if relu_y is None else relu_y.data_ptr(), 1 if dual else 0, out.data_ptr(), st)
    return (out[:C], out[C:]) if dual else out


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


def softmax_rows(S2d, scale, mul=None):
    P = torch.empty_like(S2d)
    h, st = handle_and_stream(S2d)
    h.call("gmf_softmax_rows", 0, S2d.data_ptr(), None, None if mul is None else mul.data_ptr(), P.data_ptr(), S2d.shape[0],
           S2d.shape[1], float(scale), st)
    return P


def softmax_rows_bwd(P2d, dP2d, scale, mul=None):
    dS = torch.empty_like(P2d)
    h, st = handle_and_stream(P2d)
    h.call("gmf_softmax_rows", 1, P2d.data_ptr(), dP2d.data_ptr(), None if mul is None else mul.data_ptr(), dS.data_ptr(),
           P2d.shape[0], P2d.shape[1], float(scale), st)
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
        dg2, dbn2 = colsum(dxn2, y=x1, mean=mu2, rstd=rs2, dual=True)
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
        dg1, dbn1 = colsum(dxn, y=xp, mean=mu1, rstd=rs1, dual=True)
        dxp = layernorm_bwd(dxn, xp, g1, mu1, rs1, dx_add=dx1)
        dgc, dbnc = colsum(dcn, y=cp, mean=muc, rstd=rsc, dual=True)
        dcp = layernorm_bwd(dcn, cp, gc, muc, rsc)
        grads = []
        if pe:
            # ---- LCPE (fusion_layer.py:118-128) ----
            dx = lcpe_bwd(dxp, wq_t, N)
            dc = lcpe_bwd(dcp, wc_t, T)
            tq0, dbq = colsum(dxp, y=x2, shift=0, L=N, dual=True)        # the centre tap and the bias from one pass
            tc0, dbc = colsum(dcp, y=c2, shift=0, L=T, dual=True)
            dwq = torch.stack([colsum(dxp, y=x2, shift=-1, L=N), tq0, colsum(dxp, y=x2, shift=1, L=N)], dim=1).reshape(wq_t.shape)
            dwc = torch.stack([colsum(dcp, y=c2, shift=-1, L=T), tc0, colsum(dcp, y=c2, shift=1, L=T)], dim=1).reshape(wc_t.shape)
            grads += [dwq, dbq, dwc, dbc]
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
    if getattr(layer, "depth", 0) != 0 or getattr(layer, "cross_heads", 1) != 1:
        raise NotImplementedError("gmf_amd.FusionLayer: the differentiable path covers what GMF trains (depth 0, one cross-attention head: "
                                  "PointDSC.py:29-38,92-100); latent self-attention layers / several heads are forward-only (eval())")
    return _FusionLayerTrain.apply(data, queries, bool(layer.pe), *_param_list(layer))


# =====================================================================================================================
# The rest of the encoder in training mode (PointDSC.py:10-74,77-143,175-181 with TRAIN-mode BatchNorm) and the two losses
# the reference trains with by default (libs/loss.py:67-140; config_3DMatch.py:50-52: weight_transformation = 0).
# Activations are token-major [B * N, C] rows (the reference's [B, C, N] transposed; BatchNorm1d over (B, N) per channel =
# per column over all rows).
# =====================================================================================================================
class _Linear(torch.autograd.Function):
    """y = relu?(x W^T + b (+ residual)); x [rows, in], W [out, in] (a Conv1d k = 1 weight squeezed), residual [rows, out]."""

    @staticmethod
    def forward(ctx, x, W, b, residual, relu):
        x = x.contiguous()
        W2 = W.detach().reshape(W.shape[0], -1).contiguous()
        y = gemm(x, W2, tb=True, bias=None if b is None else b.detach(), residual=None if residual is None else residual.contiguous(),
                 relu=relu)
        ctx.relu, ctx.has_b, ctx.has_r, ctx.wshape = relu, b is not None, residual is not None, W.shape
        ctx.save_for_backward(x, W2, y if relu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            g = torch.empty_like(dy)
            h, st = handle_and_stream(dy)
            h.call("gmf_relu_backward", dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), st)
            dy = g
        dW = gemm(dy, x, ta=True).reshape(ctx.wshape)
        db = colsum(dy) if ctx.has_b else None
        dx = gemm(dy, W2)
        return dx, dW, db, (dy if ctx.has_r else None), None


def linear(x, W, b=None, residual=None, relu=False):
    return _Linear.apply(x, W, b, residual, relu)


class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm1d in training mode on rows [rows, C] (+ fused ReLU); updates running_mean / running_var in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        x = x.contiguous()
        rows, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        rstd = torch.empty(C, device=x.device, dtype=torch.float32)
        h, st = handle_and_stream(x)
        h.call("gmf_batchnorm_train_forward", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
               rstd.data_ptr(), None if running_mean is None else running_mean.data_ptr(),
               None if running_var is None else running_var.data_ptr(), rows, C, float(eps), float(momentum), 1 if relu else 0, st)
        ctx.relu = relu
        ctx.save_for_backward(x, gamma.detach(), mean, rstd, y if relu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        dy = dy.contiguous()
        rows, C = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
        h, st = handle_and_stream(x)
        h.call("gmf_batchnorm_train_backward", dy.data_ptr(), x.data_ptr(), y.data_ptr() if ctx.relu else None, mean.data_ptr(),
               rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, st)
        return dx, dg, db, None, None, None, None, None


def batchnorm_train(x, bn, relu=False):
    """`bn`: an nn.BatchNorm1d whose parameters and running statistics are used / updated exactly as torch would in train()."""
    if bn.momentum is None:
        raise NotImplementedError("gmf_amd.train: BatchNorm with cumulative moving average (momentum=None) is not used by GMF")
    y = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                              bn.running_var if bn.track_running_stats else None, bn.eps, bn.momentum, relu)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return y


class _SCAttention(torch.autograd.Function):
    """message = softmax_j(compat_ij * <q_i, k_j> / sqrt(C)) V per pair (PointDSC.py:56-64).  qkv [B, N, 3C] token-major holds
    Q | K | V side by side (ONE projection GEMM produces them, one GEMM each takes their gradients back: the products below
    address the three column blocks through leading dimensions and offsets); compat [B, N, N] has no gradient (the reference
    computes it under no_grad, PointDSC.py:216-221)."""

    @staticmethod
    def forward(ctx, qkv, compat):
        qkv, compat = qkv.contiguous(), compat.contiguous()
        B, N, C3 = qkv.shape
        Cc = C3 // 3
        scale = 1.0 / (Cc ** 0.5)
        S = torch.empty((B, N, N), device=qkv.device, dtype=torch.float32)
        gemm(qkv, qkv, tb=True, out=S, m=N, n=N, k=Cc, lda=C3, ldb=C3, ldc=N, b_off=Cc, batch=B, sa=N * C3, sb=N * C3, sc=N * N)
        P = softmax_rows(S.reshape(B * N, N), scale, mul=compat.reshape(B * N, N))
        msg = torch.empty((B, N, Cc), device=qkv.device, dtype=torch.float32)
        gemm(P, qkv, out=msg, m=N, n=Cc, k=N, lda=N, ldb=C3, ldc=Cc, b_off=2 * Cc, batch=B, sa=N * N, sb=N * C3, sc=N * Cc)
        ctx.scale = scale
        ctx.save_for_backward(qkv, compat, P)
        return msg

    @staticmethod
    def backward(ctx, dmsg):
        qkv, compat, P = ctx.saved_tensors
        dmsg = dmsg.contiguous()
        B, N, C3 = qkv.shape
        Cc = C3 // 3
        dev = qkv.device
        dqkv = torch.empty_like(qkv)
        dP = torch.empty((B * N, N), device=dev, dtype=torch.float32)
        gemm(dmsg, qkv, tb=True, out=dP, m=N, n=N, k=Cc, lda=Cc, ldb=C3, ldc=N, b_off=2 * Cc, batch=B, sa=N * Cc, sb=N * C3, sc=N * N)
        gemm(P, dmsg, ta=True, out=dqkv, m=N, n=Cc, k=N, lda=N, ldb=Cc, ldc=C3, c_off=2 * Cc, batch=B, sa=N * N, sb=N * Cc,
             sc=N * C3)                                                                              # dV
        dS = softmax_rows_bwd(P, dP, ctx.scale, mul=compat.reshape(B * N, N))
        gemm(dS, qkv, out=dqkv, m=N, n=Cc, k=N, lda=N, ldb=C3, ldc=C3, b_off=Cc, batch=B, sa=N * N, sb=N * C3, sc=N * C3)     # dQ = dS K
        gemm(dS, qkv, ta=True, out=dqkv, m=N, n=Cc, k=N, lda=N, ldb=C3, ldc=C3, c_off=Cc, batch=B, sa=N * N, sb=N * C3,
             sc=N * C3)                                                                              # dK = dS^T Q
        return dqkv, None


def sc_attention(qkv, compat):
    return _SCAttention.apply(qkv, compat)


def nonlocal_block_train(block, feat, compat, image_feat, B, N):
    """NonLocalBlock.forward (PointDSC.py:40-74) on token-major rows feat [B * N, C]; returns [B * N, C]."""
    C = feat.shape[1]
    # Q | K | V from ONE product (the concatenation is a view-free torch.cat of three small parameters; autograd hands each its
    # slice of the joint gradient)
    Wqkv = torch.cat([block.projection_q.weight, block.projection_k.weight, block.projection_v.weight], dim=0)
    bqkv = torch.cat([block.projection_q.bias, block.projection_k.bias, block.projection_v.bias], dim=0)
    qkv = linear(feat, Wqkv, bqkv)
    msg = sc_attention(qkv.reshape(B, N, 3 * C), compat).reshape(B * N, C)
    fm = block.fc_message
    m1 = batchnorm_train(linear(msg, fm[0].weight, fm[0].bias), fm[1], relu=True)
    m2 = batchnorm_train(linear(m1, fm[3].weight, fm[3].bias), fm[4], relu=True)
    fused = fusion_layer_train(block.fusion_layer_2, image_feat, feat.reshape(B, N, C)).reshape(B * N, C)
    return linear(m2, fm[6].weight, fm[6].bias, residual=fused)           # message + image_feat (PointDSC.py:73)


def encoder_train(net, corr_pos, compat, p_tokens, q_tokens):
    """NonLocalNet.forward (PointDSC.py:114-143) from image TOKENS; returns corr_features [B, N, C] (= the reference's
    encoder output permuted to token-major, PointDSC.py:223-228)."""
    B, N, D = corr_pos.shape
    image_feat = fusion_layer_train(net.fusion_layer_1, p_tokens, q_tokens)            # PointDSC.py:137
    feat = linear(corr_pos.reshape(B * N, D).contiguous(), net.layer0.weight, net.layer0.bias)
    for i in range(net.num_layers):
        pc = net.blocks[f"PointCN_layer_{i}"]
        feat = batchnorm_train(linear(feat, pc[0].weight, pc[0].bias), pc[1], relu=True)
        feat = nonlocal_block_train(net.blocks[f"NonLocal_layer_{i}"], feat, compat, image_feat, B, N)
    return feat.reshape(B, N, -1)


def classifier_train(cls, feat):
    """PointDSC.classification (PointDSC.py:175-181,241): [B, N, C] -> logits [B, N]."""
    B, N, C = feat.shape
    h1 = linear(feat.reshape(B * N, C), cls[0].weight, cls[0].bias, relu=True)
    h2 = linear(h1, cls[2].weight, cls[2].bias, relu=True)
    return linear(h2, cls[4].weight, cls[4].bias).reshape(B, N)


class _Normalize(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) (PointDSC.py:229) on rows [rows, C]: LayerNorm-free row scaling, done with the primitives:
    y = x / max(||x||, 1e-12); dx = (dy - y <dy, y>) / ||x||."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        nrm = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        h, st = handle_and_stream(x)
        h.call("gmf_normalize_rows", 0, x.data_ptr(), None, nrm.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], st)
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        h, st = handle_and_stream(y)
        h.call("gmf_normalize_rows", 1, y.data_ptr(), dy.data_ptr(), nrm.data_ptr(), dx.data_ptr(), y.shape[0], y.shape[1], st)
        return dx


def normalize_rows(x):
    return _Normalize.apply(x)


class _SimilarityMatrix(torch.autograd.Function):
    """M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1), zero diagonal (PointDSC.py:231-234), dense [B, N, N], differentiable with
    respect to Fn and sigma for any upstream gradient (gmf_similarity_backward)."""

    @staticmethod
    def forward(ctx, feat_n, sigma, sigma_value):
        f = feat_n.contiguous()
        B, N, _ = f.shape
        on_dev = sigma_value is SIGMA_ON_DEVICE          # [r5] no host read: the kernels take sigma from the parameter's own memory
        sig = 1.0 if on_dev else (float(sigma) if sigma_value is None else float(sigma_value))     # (a caller that already read the scalar passes it)
        M = torch.empty((B, N, N), device=f.device, dtype=torch.float32)
        h, st = handle_and_stream(f)
        with sigma_device(h, sigma if on_dev else None):
            h.call("gmf_similarity_matrix", f.data_ptr(), B, N, sig, M.data_ptr(), N, st)
        ctx.sig, ctx.sigma_is_tensor, ctx.sigma_dev = sig, torch.is_tensor(sigma), (sigma if on_dev else None)
        ctx.save_for_backward(f)
        return M

    @staticmethod
    def backward(ctx, dM):
        (f,) = ctx.saved_tensors
        B, N, _ = f.shape
        dM = dM.contiguous()
        dF = torch.empty_like(f)
        dsig = torch.empty(1, device=f.device, dtype=torch.float32)
        h, st = handle_and_stream(f)
        with sigma_device(h, ctx.sigma_dev):
            h.call("gmf_similarity_backward", f.data_ptr(), dM.data_ptr(), B, N, ctx.sig, dF.data_ptr(), dsig.data_ptr(), st)
        return dF, dsig if ctx.sigma_is_tensor else None, None


def similarity_matrix_train(feat_n, sigma, sigma_value=None):
    return _SimilarityMatrix.apply(feat_n, sigma, sigma_value)


class _ClassificationLossFn(torch.autograd.Function):
    """The loss value of ClassificationLoss (libs/loss.py:67-93) with its gradient with respect to the logits."""

    @staticmethod
    def forward(ctx, pred, gt, weight, balanced):
        pred = pred.contiguous()
        out = torch.empty(6, device=pred.device, dtype=torch.float32)
        h, st = handle_and_stream(pred)
        h.call("gmf_classification_loss", pred.data_ptr(), gt.data_ptr(), None if weight is None else weight.data_ptr(),
               pred.shape[0], pred.shape[1], 1 if balanced else 0, out.data_ptr(), st)
        ctx.balanced = balanced
        ctx.save_for_backward(pred, gt, weight if weight is not None else pred.new_empty(0))
        ctx.has_w = weight is not None
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, dloss, _dstats):
        pred, gt, weight = ctx.saved_tensors
        d = torch.empty_like(pred)
        h, st = handle_and_stream(pred)
        h.call("gmf_classification_backward", pred.data_ptr(), gt.data_ptr(), weight.data_ptr() if ctx.has_w else None,
               pred.shape[0], pred.shape[1], 1 if ctx.balanced else 0, d.data_ptr(), st)
        return dloss * d, None, None, None


def classification_loss_train(pred, gt, weight, balanced):
    return _ClassificationLossFn.apply(pred, gt, weight, balanced)


class _SpectralMatchingDense(torch.autograd.Function):
    """SpectralMatchingLoss(M, gt) (libs/loss.py:116-140) on a dense M with its gradient dL/dM."""

    @staticmethod
    def forward(ctx, M, gt, balanced):
        N = M.shape[1]
        if not (M.stride(2) == 1 and M.stride(1) >= N and M.stride(0) == N * M.stride(1)):
            M = M.contiguous()
        out = torch.empty(1, device=M.device, dtype=torch.float32)
        h, st = handle_and_stream(M)
        h.call("gmf_spectral_matching_loss", M.data_ptr(), M.stride(1), gt.data_ptr(), M.shape[0], N, 1 if balanced else 0,
               out.data_ptr(), st)
        ctx.balanced = balanced
        ctx.save_for_backward(M, gt)
        return out[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        M, gt = ctx.saved_tensors
        B, N = M.shape[0], M.shape[1]
        dM = torch.empty((B, N, N), device=M.device, dtype=torch.float32)
        h, st = handle_and_stream(M)
        h.call("gmf_spectral_matching_dense_backward", M.data_ptr(), M.stride(1), gt.data_ptr(), B, N, 1 if ctx.balanced else 0,
               dM.data_ptr(), st)
        return dloss * dM, None, None


def spectral_matching_loss_train(M, gt, balanced):
    return _SpectralMatchingDense.apply(M, gt, balanced)


def compat_dense(src, tgt, sigma_d):
    """[B, N, N] spatial-consistency matrix of PointDSC.py:216-221 (no gradient, as in the reference)."""
    B, N, _ = src.shape
    out = torch.empty((B, N, N), device=src.device, dtype=torch.float32)
    h, st = handle_and_stream(src)
    h.call("gmf_compat_dense", src.data_ptr(), tgt.data_ptr(), B, N, float(sigma_d), out.data_ptr(), st)
    return out


class _PoseHeadTrain(torch.autograd.Function):
    """final_trans of the non-test forward (PointDSC.py:246-252,304-425): top-S seeds by confidence, kNN in feature space,
    compatibility matrix + power iteration + weighted Kabsch per seed, the hypothesis with most inliers.  Forward =
    gmf_pose_head; backward = gmf_pose_head_backward (only the best seed of each pair carries gradient: to the k neighbour
    rows of the unit features and to sigma)."""

    @staticmethod
    def forward(ctx, feat_n, sigma, model, src, tgt, logits, sigmas):
        f = feat_n.contiguous()
        on_dev = sigmas is not None and sigmas[0] is SIGMA_ON_DEVICE
        if on_dev:
            sigmas = (1.0, sigmas[1])                    # (placeholder: the kernels read sigma from the parameter's memory)
        h, _ = handle_and_stream(f)
        with sigma_device(h, sigma if on_dev else None):
            final_T, _, aux = model.pose_head(f, src, tgt, logits, False, return_aux=True, sigmas=sigmas)
        ctx.model, ctx.sigmas, ctx.sigma_is_tensor, ctx.sigma_dev = model, sigmas, torch.is_tensor(sigma), (sigma if on_dev else None)
        ctx.save_for_backward(f, src, tgt, aux["knn_idx"], aux["fitness"])
        return final_T

    @staticmethod
    def backward(ctx, dT):
        f, src, tgt, knn_idx, fitness = ctx.saved_tensors
        B, N, _ = f.shape
        pp, _, _ = ctx.model._pose_params(N, False, ctx.sigmas)
        dT = dT.contiguous()
        dF = torch.empty_like(f)
        dsig = torch.empty(B, device=f.device, dtype=torch.float32)
        h, st = handle_and_stream(f)
        with sigma_device(h, ctx.sigma_dev):
            h.call("gmf_pose_head_backward", pp, f.data_ptr(), src.data_ptr(), tgt.data_ptr(), knn_idx.data_ptr(), fitness.data_ptr(),
                   dT.data_ptr(), B, N, dF.data_ptr(), dsig.data_ptr(), st)
        return dF, (dsig.sum().reshape(1) if ctx.sigma_is_tensor else None), None, None, None, None, None


def pose_head_train(model, feat_n, sigma, src, tgt, logits, sigmas):
    return _PoseHeadTrain.apply(feat_n, sigma, model, src, tgt, logits, sigmas)


class GraphedTrainingStep:
    """One training step of `model` (a gmf_amd.PointDSC in train() mode) - forward, loss, backward, optimizer step - captured ONCE as a
    HIP graph and replayed (libs/trainer.py:131-166 is the loop this stands in for).  An eager step is ~1 900 kernel launches from
    ~320 C calls; the replay is one launch from Python.  It is NOT shorter than a well-fed eager step: the kernels are dependent and
    sum to ~24 ms at 16 x 1000 - the graph removes the host from the loop, not time from the device.

    What makes the step capturable: `model.sigma_on_device = True` (the learnable sigma is read by the kernels from the parameter's
    own memory, gmf_set_sigma_device - the eager step reads it to the host once per step), a `loss_fn(result, batch) -> 0-dim device
    tensor` that makes no host read (`ClassificationLoss(host_stats=False)`), and an optimizer whose step is capturable
    (`torch.optim.Adam(..., capturable=True, fused=True)` - fused: torch's multi-tensor Adam under capture issues three broadcast
    divisions per parameter, ~950 launches and 4 ms of a 27 ms step at 16 x 1000).  `warmup` eager steps run first, on a side stream: they size the library's workspace
    and the allocator's pools, and they are real optimizer steps.  Shapes are fixed by the example batch; `__call__(batch)` copies
    the tensors of `batch` into the captured inputs and replays."""

    def __init__(self, model, optimizer, loss_fn, batch, warmup: int = 3):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        model.sigma_on_device = True
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}

        def step():
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn(model(self.static), self.static)
            loss.backward()
            optimizer.step()
            return loss
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def __call__(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v)
        self.graph.replay()
        return self.loss

