"""weighted_procrustes drop-in for the DGR plugin surface
(reference: GMF_DeepGlobalRegistration/*/core/registration.py:91-113)."""
from __future__ import annotations

from typing import Sequence

import torch

from ._util import handle_and_stream, require_cuda_f32


def weighted_procrustes_batched(X, Y, w, offsets: Sequence[int], eps):
    """B ragged problems in one launch: X,Y [sum N,3], w [sum N] or [sum N,1], offsets (B+1 ints) -> R [B,3,3], t [B,3]."""
    X = require_cuda_f32(X, "X").contiguous()
    Y = require_cuda_f32(Y, "Y").contiguous()
    w = require_cuda_f32(w, "w").contiguous().reshape(-1)
    if X.shape != Y.shape or X.dim() != 2 or X.shape[1] != 3 or w.numel() != X.shape[0]:
        raise RuntimeError("gmf_amd.weighted_procrustes: expected X,Y [N,3] and w [N]")
    off = torch.as_tensor(list(offsets), dtype=torch.int32)
    B = off.numel() - 1
    if B < 1 or int(off[0]) != 0 or int(off[-1]) != X.shape[0] or bool((off[1:] <= off[:-1]).any()):
        raise RuntimeError("gmf_amd.weighted_procrustes: offsets must be increasing, start at 0 and end at N")
    off = off.to(X.device)
    R = torch.empty((B, 3, 3), device=X.device, dtype=torch.float32)
    t = torch.empty((B, 3), device=X.device, dtype=torch.float32)
    h, st = handle_and_stream(X)
    h.call("gmf_weighted_procrustes", X.data_ptr(), Y.data_ptr(), w.data_ptr(), off.data_ptr(), B, float(eps),
           R.data_ptr(), t.data_ptr(), st)
    return R, t


def weighted_procrustes(X, Y, w, eps):
    """X,Y [N,3], w [N] or [N,1] -> (R [3,3], t [3]) fp32 (registration.py:91-113).

    The reference moves Sxy to the CPU for an fp64 SVD (registration.py:105-106); here the reduction and
    the fp64 SVD run in one HIP kernel and the result stays on the device."""
    assert len(X) == len(Y)
    R, t = weighted_procrustes_batched(X, Y, w, [0, X.shape[0]], eps)
    return R[0], t[0]
