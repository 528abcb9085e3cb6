"""DGR plugin surface: weighted_procrustes and GlobalRegistration drop-ins
(reference: GMF_DeepGlobalRegistration/*/core/registration.py:91-113, 135-194)."""
from __future__ import annotations

from typing import Sequence

import torch

from ._util import handle_and_stream, require_cuda_f32


import collections
import os
import threading

# (device, offsets) -> (int32 device tensor, largest problem): an LRU of the 64 most recent offset lists behind a lock (callers on
# several threads share it).  Evaluation loops that repeat a handful of shapes hit it; a loop whose offsets change on every
# call pays the upload each time - through a pinned staging buffer and an asynchronous copy, not a pageable synchronous one.
_OFFSETS_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_OFFSETS_LOCK = threading.Lock()
_OFFSETS_CAP = 64


def _device_offsets(offsets, n_rows: int, device, what: str):
    """The row offsets of a ragged batch as an int32 device tensor, and the largest problem.  A Python sequence is checked
    here (increasing, from 0 to n_rows) without a device round trip and its device copy is kept for the next call with the
    same offsets (a list met for the first time travels through a pinned buffer, asynchronously); an int32 tensor already on the device
    is taken as it is (the caller vouches for it: only its end is not read back)."""
    if isinstance(offsets, torch.Tensor) and offsets.is_cuda:
        if offsets.dtype != torch.int32 or offsets.dim() != 1 or offsets.numel() < 2 or not offsets.is_contiguous() \
                or offsets.device != device:
            raise RuntimeError(f"gmf_amd.{what}: device offsets must be a contiguous int32 vector of B + 1 entries on the points' device")
        if os.environ.get("GMF_DEBUG_OFFSETS"):          # (debugging aid: one device read-back of the whole vector)
            off_h = offsets.cpu().tolist()
            if off_h[0] != 0 or off_h[-1] != n_rows or any(b <= a for a, b in zip(off_h, off_h[1:])):
                raise RuntimeError(f"gmf_amd.{what}: device offsets must be increasing, start at 0 and end at N")
        return offsets, n_rows
    off = tuple(int(o) for o in offsets)
    if len(off) < 2 or off[0] != 0 or off[-1] != n_rows or any(b <= a for a, b in zip(off, off[1:])):
        raise RuntimeError(f"gmf_amd.{what}: offsets must be increasing, start at 0 and end at N")
    key = (device, off)
    with _OFFSETS_LOCK:
        hit = _OFFSETS_CACHE.get(key)
        if hit is not None:
            _OFFSETS_CACHE.move_to_end(key)
            if hit[3] is not None:                   # the upload may still be in flight on ANOTHER stream: order this one behind it
                if hit[3].query():
                    hit[3] = None
                else:
                    torch.cuda.current_stream(device).wait_event(hit[3])
            return hit[0], hit[1]
    # not cached: a pinned host copy and an asynchronous upload on the current stream (ordered before the solve that follows)
    host = torch.tensor(off, dtype=torch.int32).pin_memory()
    dev_off = host.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    entry = [dev_off, max(b - a for a, b in zip(off, off[1:])), host, ev]     # (the pinned source lives as long as the entry)
    with _OFFSETS_LOCK:
        _OFFSETS_CACHE[key] = entry
        while len(_OFFSETS_CACHE) > _OFFSETS_CAP:
            _OFFSETS_CACHE.popitem(last=False)
    return entry[0], entry[1]


class _WeightedProcrustes(torch.autograd.Function):
    """(R, t) of the batched solve, differentiable with respect to the weights (gmf_weighted_procrustes_backward: the DGR
    trainer back-propagates its pose losses through this solve into the inlier network, core/trainer.py:594-614)."""

    @staticmethod
    def forward(ctx, w, X, Y, off, eps):
        B = off.numel() - 1
        out = torch.empty(B * 12, device=X.device, dtype=torch.float32)          # one allocation: R | t
        R, t = out[:B * 9].view(B, 3, 3), out[B * 9:].view(B, 3)
        h, st = handle_and_stream(X)
        h.call("gmf_weighted_procrustes", X.data_ptr(), Y.data_ptr(), w.data_ptr(), off.data_ptr(), B, float(eps),
               R.data_ptr(), t.data_ptr(), st)
        if ctx is not None:
            ctx.eps = float(eps)
            ctx.save_for_backward(w, X, Y, off)
        return R, t

    @staticmethod
    def backward(ctx, dR, dt):
        w, X, Y, off = ctx.saved_tensors
        B = off.numel() - 1
        dR = torch.zeros((B, 3, 3), device=X.device) if dR is None else dR.contiguous()
        dt = torch.zeros((B, 3), device=X.device) if dt is None else dt.contiguous()
        dw = torch.empty_like(w)
        h, st = handle_and_stream(X)
        h.call("gmf_weighted_procrustes_backward", X.data_ptr(), Y.data_ptr(), w.data_ptr(), off.data_ptr(), B, ctx.eps,
               dR.data_ptr(), dt.data_ptr(), dw.data_ptr(), st)
        return dw, None, None, None, None


def weighted_procrustes_batched(X, Y, w, offsets: Sequence[int], eps):
    """B ragged problems in one launch: X,Y [sum N,3], w [sum N] or [sum N,1], offsets (B+1 ints) -> R [B,3,3], t [B,3].
    With autograd enabled and `w` requiring grad the result carries the gradient with respect to w (X, Y: none, as in the
    DGR trainer, which sets requires_grad = False on them)."""
    X = require_cuda_f32(X, "X").contiguous()
    Y = require_cuda_f32(Y, "Y").contiguous()
    w = require_cuda_f32(w, "w").contiguous().reshape(-1)
    if X.shape != Y.shape or X.dim() != 2 or X.shape[1] != 3 or w.numel() != X.shape[0]:
        raise RuntimeError("gmf_amd.weighted_procrustes: expected X,Y [N,3] and w [N]")
    if torch.is_grad_enabled() and (X.requires_grad or Y.requires_grad):
        raise RuntimeError("gmf_amd.weighted_procrustes: gradients with respect to X / Y are not implemented (only w)")
    off, _ = _device_offsets(offsets, X.shape[0], X.device, "weighted_procrustes")
    if not (torch.is_grad_enabled() and w.requires_grad):
        return _WeightedProcrustes.forward(None, w, X, Y, off, eps)
    return _WeightedProcrustes.apply(w, X.detach(), Y.detach(), off, eps)


def weighted_procrustes(X, Y, w, eps):
    """X,Y [N,3], w [N] or [N,1] -> (R [3,3], t [3]) fp32 (registration.py:91-113).

    The reference moves Sxy to the CPU for an fp64 SVD (registration.py:105-106); here the reduction and
    the fp64 SVD run in one HIP kernel and the result stays on the device."""
    assert len(X) == len(Y)
    R, t = weighted_procrustes_batched(X, Y, w, [0, X.shape[0]], eps)
    return R[0], t[0]


_F32_EPS = 1.1920928955078125e-07      # np.finfo(np.float32).eps, the default eps of HighDimSmoothL1Loss (core/loss.py:44)


def global_registration_batched(points, trans_points, weights, offsets: Sequence[int], max_iter=1000, max_break_count=20,
                                break_threshold_ratio=1e-5, quantization_size=1, eps=_F32_EPS):
    """B ragged problems in one launch (one persistent workgroup each): points, trans_points [sum N,3], weights [sum N]
    / [sum N,1] or None, offsets (B+1 ints) -> R [B,3,3], t [B,3], stats [B,3] = (iterations, loss, break_count)."""
    X = require_cuda_f32(points, "points").contiguous()
    Y = require_cuda_f32(trans_points, "trans_points").contiguous()
    if X.shape != Y.shape or X.dim() != 2 or X.shape[1] != 3:
        raise RuntimeError("gmf_amd.GlobalRegistration: expected points, trans_points [N,3]")
    w = None
    if weights is not None:
        w = require_cuda_f32(weights, "weights").contiguous().reshape(-1)
        if w.numel() != X.shape[0]:
            raise RuntimeError("gmf_amd.GlobalRegistration: weights must hold one value per point")
    off, max_points = _device_offsets(offsets, X.shape[0], X.device, "GlobalRegistration")
    B = off.numel() - 1
    R = torch.empty((B, 3, 3), device=X.device, dtype=torch.float32)
    t = torch.empty((B, 3), device=X.device, dtype=torch.float32)
    stats = torch.empty((B, 3), device=X.device, dtype=torch.float32)
    h, st = handle_and_stream(X)
    h.call("gmf_global_registration", X.data_ptr(), Y.data_ptr(), None if w is None else w.data_ptr(), off.data_ptr(), B,
           float(eps), float(quantization_size), int(max_iter), int(max_break_count), float(break_threshold_ratio),
           R.data_ptr(), t.data_ptr(), stats.data_ptr(), max_points, st)
    return R, t, stats


def GlobalRegistration(points, trans_points, weights=None, max_iter=1000, verbose=False, stat_freq=20,
                       max_break_count=20, break_threshold_ratio=1e-5, loss_fn=None, quantization_size=1):
    """Drop-in for core/registration.py:135-194: (R [3,3], t [3], {'iterations', 'loss', 'break_count'}).

    numpy inputs are accepted as in the reference (:145-149) and moved to the current HIP device; `verbose` /
    `stat_freq` only print in the reference and are ignored; a custom `loss_fn` is not supported (the reference's own
    callers never pass one) and raises."""
    import numpy as np
    if loss_fn is not None:
        raise NotImplementedError("gmf_amd.GlobalRegistration: only the default HighDimSmoothL1Loss is implemented")
    dev = None
    for v in (points, trans_points, weights):
        if isinstance(v, torch.Tensor) and v.is_cuda:
            dev = v.device
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device())

    def _t(v):
        if v is None:
            return None
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.detach().float().to(dev)

    P, Q, W = _t(points), _t(trans_points), _t(weights)
    R, t, stats = global_registration_batched(P, Q, W, [0, P.shape[0]], max_iter, max_break_count,
                                              break_threshold_ratio, quantization_size)
    s = stats[0].tolist()
    return R[0], t[0], {"iterations": int(s[0]), "loss": s[1], "break_count": int(s[2])}


def argmin_se3_squared_dist(X, Y):
    """Drop-in for core/registration.py:67-88: the unweighted Kabsch / Umeyama solve, (R [3,3], t [3]) of
    argmin sum |R x_i + t - y_i|^2.  It is `weighted_procrustes` with unit weights and eps = 0 (w / (sum |w| + 0) = 1 / N, the
    reference's `/ len(X)`); the 3 x 3 SVD runs on the device in fp64 (the reference's runs in X's own precision)."""
    assert len(X) == len(Y)
    return weighted_procrustes(X, Y, torch.ones(X.shape[0], device=X.device, dtype=torch.float32), 0.0)


def ortho2rotation(poses):
    """[B, 6] -> [B, 3, 3]: Gram-Schmidt of the two 3-vectors, third column their cross product (core/registration.py:16-64; Zhou et
    al.'s continuous 6-D rotation).  Plain torch - it is the parameterisation `Transformation` differentiates through; the
    refinement loop itself (GlobalRegistration) evaluates it inside its persistent kernel."""
    def unit(v):
        return v / torch.clamp(torch.sqrt((v ** 2).sum(1, keepdim=True)), min=1e-8)
    x_raw, y_raw = poses[:, 0:3], poses[:, 3:6]
    x = unit(x_raw)
    proj = ((x * y_raw).sum(1, keepdim=True) / torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8)) * x
    y = unit(y_raw - proj)
    z = torch.cross(x, y, dim=1)
    return torch.stack((x, y, z), dim=2)


class Transformation(torch.nn.Module):
    """Drop-in for core/registration.py:116-132: the 6-D rotation + translation parameters GlobalRegistration optimises, with the
    reference's initialisation (the first two COLUMNS of R_init) and forward (points @ R^T + t)."""

    def __init__(self, R_init=None, t_init=None):
        super().__init__()
        rot_init = torch.rand(1, 6)
        trans_init = torch.zeros(1, 3)
        if R_init is not None:
            rot_init[0, :3] = R_init[:, 0]
            rot_init[0, 3:] = R_init[:, 1]
        if t_init is not None:
            trans_init[0] = t_init
        self.rot6d = torch.nn.Parameter(rot_init)
        self.trans = torch.nn.Parameter(trans_init)

    def forward(self, points):
        return points @ ortho2rotation(self.rot6d)[0].t() + self.trans

