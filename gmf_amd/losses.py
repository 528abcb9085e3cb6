"""The three loss / metric modules of the PointDSC plugin surface (reference: GMF_PointDSC/libs/loss.py; callers
libs/trainer.py:131-143 - the training step - and :194-262, evaluate()): the reference's class names, constructor arguments,
call signatures and return values, each one launch sequence of libgmf_hip.so over the C ABI.  All three carry their
gradients: `ClassificationLoss(pred, gt)` with respect to the logits, `SpectralMatchingLoss(M, gt)`
with respect to M (and `SpectralMatchingLoss.from_features` - the fused similarity matrix + loss, one launch each way - with
respect to the unit features and sigma), `TransformationLoss` with respect to `trans`.  The per-element `weight` of
ClassificationLoss is a constant and must not require grad."""
from __future__ import annotations

import torch
import torch.nn as nn

from ._util import handle_and_stream, require_cuda_f32


def _no_grad(*tensors):
    for t in tensors:
        if t is not None and t.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("gmf_amd.losses: this argument is a constant of the loss (no gradient is defined for it); "
                               "detach it or call under torch.no_grad()")


def similarity_matrix(feat_n: torch.Tensor, sigma: float, contiguous: bool = False) -> torch.Tensor:
    """PointDSC.py:231-234: feat_n [B,N,128] unit rows -> M [B,N,N] = clamp(1 - (1 - Fn Fn^T)/sigma^2, 0, 1), diag 0.

    By default M is a [B,N,N] view whose rows are padded to a multiple of 32 floats (stride(1) = ceil(N/32)*32): every
    row piece the kernel stores is then a whole 128-byte line, which nearly doubles the store rate at N = 5000.  All
    torch operations accept the view; `contiguous=True` writes the dense layout instead."""
    f = require_cuda_f32(feat_n, "feat_n").contiguous()
    if f.dim() != 3 or f.shape[2] != 128:
        raise RuntimeError("gmf_amd.similarity_matrix: expected feat_n [B,N,128]")
    B, N, _ = f.shape
    ldm = N if contiguous else (N + 31) // 32 * 32
    buf = torch.empty((B, N, ldm), device=f.device, dtype=torch.float32)
    h, st = handle_and_stream(f)
    h.call("gmf_similarity_matrix", f.data_ptr(), B, N, float(sigma), buf.data_ptr(), ldm, st)
    return buf if ldm == N else buf[:, :, :N]


class ClassificationLoss(nn.Module):
    """libs/loss.py:67-113.  forward(pred [bs,N] logits, gt [bs,N] 0/1, weight=None) -> dict with 'loss' (0-dim device
    tensor) and the floats 'precision', 'recall', 'f1' (pair 0 only, as the reference), 'logit_true', 'logit_false'."""

    def __init__(self, balanced=True, host_stats=True):
        super().__init__()
        self.balanced = balanced
        # [r5] host_stats = False: no device-to-host read - the dict carries 'loss' and 'stats' (a device tensor: loss, precision,
        # recall, f1, logit_true, logit_false), which is what a graph-captured training step needs (GraphedTrainingStep)
        self.host_stats = host_stats

    def forward(self, pred, gt, weight=None):
        _no_grad(weight)
        pred = require_cuda_f32(pred, "pred").contiguous()
        gt = gt.to(device=pred.device, dtype=torch.float32).contiguous()
        if pred.dim() != 2 or gt.shape != pred.shape:
            raise RuntimeError("gmf_amd.ClassificationLoss: expected pred and gt of shape [bs, num_corr]")
        w = None
        if weight is not None:
            w = require_cuda_f32(weight, "weight").expand_as(pred).contiguous()
        if pred.requires_grad and torch.is_grad_enabled():      # training: the loss carries d loss / d logits (libs/trainer.py:134)
            from .train import classification_loss_train
            loss, out = classification_loss_train(pred, gt, w, self.balanced)
            if not self.host_stats:
                return {"loss": loss, "stats": out}
            host = out.cpu()
            return {"loss": loss, "precision": float(host[1]), "recall": float(host[2]), "f1": float(host[3]),
                    "logit_true": float(host[4]), "logit_false": float(host[5])}
        out = torch.empty(6, device=pred.device, dtype=torch.float32)
        h, st = handle_and_stream(pred)
        h.call("gmf_classification_loss", pred.data_ptr(), gt.data_ptr(), None if w is None else w.data_ptr(),
               pred.shape[0], pred.shape[1], 1 if self.balanced else 0, out.data_ptr(), st)
        host = out.cpu()          # the reference also leaves the device here (sklearn scores, loss.py:98-104)
        return {"loss": out[0], "precision": float(host[1]), "recall": float(host[2]), "f1": float(host[3]),
                "logit_true": float(host[4]), "logit_false": float(host[5])}


class _SpectralMatchingFromFeatures(torch.autograd.Function):
    """loss = SpectralMatchingLoss(M(feat_n, sigma), gt) with M = clamp(1 - (1 - Fn Fn^T)/sigma^2, 0, 1), zero diagonal
    (PointDSC.py:231-234, libs/loss.py:116-140).  forward: gmf_spectral_matching_loss_fused; backward:
    gmf_spectral_matching_backward (d loss / d feat_n [B,N,128], d loss / d sigma) - M is written in neither."""

    @staticmethod
    def forward(ctx, feat_n, sigma, gt, balanced):
        f = require_cuda_f32(feat_n, "feat_n").contiguous()
        out = torch.empty(1, device=f.device, dtype=torch.float32)
        sig = float(sigma)                       # one host read of the scalar parameter (the reference's `self.sigma ** 2`)
        h, st = handle_and_stream(f)
        h.call("gmf_spectral_matching_loss_fused", f.data_ptr(), gt.data_ptr(), f.shape[0], f.shape[1], sig,
               1 if balanced else 0, out.data_ptr(), st)
        ctx.save_for_backward(f, gt)
        ctx.sig, ctx.balanced, ctx.sigma_is_tensor = sig, balanced, torch.is_tensor(sigma)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        f, gt = ctx.saved_tensors
        dF = torch.empty_like(f)
        dsig = torch.empty(1, device=f.device, dtype=torch.float32)
        h, st = handle_and_stream(f)
        h.call("gmf_spectral_matching_backward", f.data_ptr(), gt.data_ptr(), f.shape[0], f.shape[1], ctx.sig,
               1 if ctx.balanced else 0, dF.data_ptr(), dsig.data_ptr(), st)
        return grad_out * dF, (grad_out * dsig).reshape(1) if ctx.sigma_is_tensor else None, None, None


class SpectralMatchingLoss(nn.Module):
    """libs/loss.py:116-140.  forward(M [bs,N,N], gt_labels [bs,N]) -> 0-dim device tensor.

    `from_features(feat_n, sigma, gt_labels)` gives the same value without M ever being written (the N x N product is
    reduced tile by tile): the form to use when only the loss, not M itself, is wanted."""

    def __init__(self, balanced=True):
        super().__init__()
        self.balanced = balanced

    def forward(self, M, gt_labels):
        M = require_cuda_f32(M, "M")
        if M.dim() != 3 or M.shape[1] != M.shape[2] or tuple(gt_labels.shape) != tuple(M.shape[:2]):
            raise RuntimeError("gmf_amd.SpectralMatchingLoss: expected M [bs,N,N] and gt_labels [bs,N]")
        N = M.shape[1]
        if M.requires_grad and torch.is_grad_enabled():         # training: the loss carries dL/dM (libs/trainer.py:137)
            from .train import spectral_matching_loss_train
            return spectral_matching_loss_train(M, gt_labels.to(device=M.device, dtype=torch.float32).contiguous(), self.balanced)
        if not (M.stride(2) == 1 and M.stride(1) >= N and M.stride(0) == N * M.stride(1)):   # row-padded views pass
            M = M.contiguous()
        gt = gt_labels.to(device=M.device, dtype=torch.float32).contiguous()
        out = torch.empty(1, device=M.device, dtype=torch.float32)
        h, st = handle_and_stream(M)
        h.call("gmf_spectral_matching_loss", M.data_ptr(), M.stride(1), gt.data_ptr(), M.shape[0], N,
               1 if self.balanced else 0, out.data_ptr(), st)
        return out[0]

    def from_features(self, feat_n, sigma, gt_labels):
        """Differentiable: with grad enabled and `feat_n` (or a tensor `sigma` of one element, e.g. PointDSC.sigma) requiring
        grad, the result carries the backward of the whole M + loss computation (one HIP launch, M never written)."""
        f = require_cuda_f32(feat_n, "feat_n")
        gt = gt_labels.to(device=f.device, dtype=torch.float32).contiguous()
        if f.dim() != 3 or f.shape[2] != 128 or gt.shape != f.shape[:2]:
            raise RuntimeError("gmf_amd.SpectralMatchingLoss.from_features: expected feat_n [bs,N,128] and gt_labels [bs,N]")
        needs_grad = torch.is_grad_enabled() and (f.requires_grad or (torch.is_tensor(sigma) and sigma.requires_grad))
        if needs_grad:
            if torch.is_tensor(sigma) and sigma.numel() != 1:
                raise RuntimeError("gmf_amd.SpectralMatchingLoss.from_features: sigma must be a scalar or a one-element tensor")
            return _SpectralMatchingFromFeatures.apply(f, sigma, gt, self.balanced)
        f = f.contiguous()
        out = torch.empty(1, device=f.device, dtype=torch.float32)
        h, st = handle_and_stream(f)
        h.call("gmf_spectral_matching_loss_fused", f.data_ptr(), gt.data_ptr(), f.shape[0], f.shape[1], float(sigma),
               1 if self.balanced else 0, out.data_ptr(), st)
        return out[0]


class _TransformationLossFn(torch.autograd.Function):
    """The five outputs of TransformationLoss (gmf_transformation_loss); element 0, the loss, carries its gradient with
    respect to `trans` (gmf_transformation_loss_backward)."""

    @staticmethod
    def forward(ctx, trans, gt_trans, src, tgt, probs, re_thre, te_thre):
        trans = trans.contiguous()
        bs, N = probs.shape
        out = torch.empty(5, device=trans.device, dtype=torch.float32)
        h, st = handle_and_stream(trans)
        h.call("gmf_transformation_loss", trans.data_ptr(), gt_trans.data_ptr(), src.data_ptr(), tgt.data_ptr(),
               probs.data_ptr(), bs, N, re_thre, te_thre, out.data_ptr(), st)
        ctx.save_for_backward(trans, src, tgt, probs)
        return out

    @staticmethod
    def backward(ctx, dout):
        trans, src, tgt, probs = ctx.saved_tensors
        bs, N = probs.shape
        g = torch.empty_like(trans)
        h, st = handle_and_stream(trans)
        h.call("gmf_transformation_loss_backward", trans.data_ptr(), src.data_ptr(), tgt.data_ptr(), probs.data_ptr(), bs, N,
               g.data_ptr(), st)
        return g * dout[0], None, None, None, None, None, None


class TransformationLoss(nn.Module):
    """libs/loss.py:12-64.  forward(trans, gt_trans [bs,4,4], src_keypts, tgt_keypts [bs,N,3], probs [bs,N]) ->
    (loss, recall %, RE deg, TE cm, RMSE); loss, RE, TE, RMSE are 0-dim device tensors, recall a float, as the
    reference returns them."""

    def __init__(self, re_thre=15, te_thre=30):
        super().__init__()
        self.re_thre = re_thre
        self.te_thre = te_thre

    def forward(self, trans, gt_trans, src_keypts, tgt_keypts, probs):
        trans = require_cuda_f32(trans, "trans")
        dev = trans.device
        gt_trans = gt_trans.to(device=dev, dtype=torch.float32).contiguous()
        src = require_cuda_f32(src_keypts, "src_keypts").contiguous()
        tgt = require_cuda_f32(tgt_keypts, "tgt_keypts").contiguous()
        probs = require_cuda_f32(probs, "probs").detach().contiguous()       # (used as a mask only: probs > 0, loss.py:57)
        bs, N = probs.shape
        if trans.shape != (bs, 4, 4) or gt_trans.shape != (bs, 4, 4) or src.shape != (bs, N, 3) or tgt.shape != (bs, N, 3):
            raise RuntimeError("gmf_amd.TransformationLoss: expected trans, gt_trans [bs,4,4], keypts [bs,N,3], probs [bs,N]")
        out = _TransformationLossFn.apply(trans, gt_trans, src, tgt, probs, float(self.re_thre), float(self.te_thre))
        o = out.detach()
        return out[0], float(o[1]), o[2], o[3], o[4]
