"""rigid_transform_3d / knn drop-ins (reference: GMF_PointDSC/models/common.py:10-75)."""
from __future__ import annotations

import torch

from ._util import handle_and_stream, require_cuda_f32


def rigid_transform_3d(A, B, weights=None, weight_threshold=0):
    """Weighted Kabsch: A,B [bs,k,3], weights [bs,k] -> T [bs,4,4] (common.py:10-50).

    The reference ships every 3x3 covariance to the CPU for torch.svd (common.py:40-41); here the whole
    solve, SVD included, is one HIP kernel.  Like the reference, weights below the threshold are zeroed
    IN PLACE in the caller's tensor (common.py:24).
    """
    A = require_cuda_f32(A, "A").contiguous()
    B = require_cuda_f32(B, "B").contiguous()
    if A.shape != B.shape or A.dim() != 3 or A.shape[2] != 3:
        raise RuntimeError(f"gmf_amd.rigid_transform_3d: expected A,B [bs,k,3], got {tuple(A.shape)} / {tuple(B.shape)}")
    bs, k = A.shape[0], A.shape[1]
    w = None
    if weights is not None:
        require_cuda_f32(weights, "weights")
        weights[weights < weight_threshold] = 0
        w = weights.contiguous()
    T = torch.empty((bs, 4, 4), device=A.device, dtype=torch.float32)
    if bs == 0:
        return T
    if k == 0:
        raise RuntimeError("gmf_amd.rigid_transform_3d: empty point set")
    h, st = handle_and_stream(A)
    h.call("gmf_procrustes_batched", A.data_ptr(), B.data_ptr(), None if w is None else w.data_ptr(), bs, k,
           float(weight_threshold), T.data_ptr(), st)
    return T


def knn(x, k, ignore_self=False, normalized=True):
    """Indices of the k nearest rows under 2 - 2 x x^T (common.py:53-75): x [bs,N,C] -> [bs,N,k] int64.

    `normalized=True` (unit rows - what GMF feeds, PointDSC.py:229,327) with either value of `ignore_self`: the reference takes
    top-(k+1) and drops rank 0 (`ignore_self=True`) or top-k including rank 0, which for unit rows is the row itself (distance 0;
    [r5] returned here as the row's own index followed by its k - 1 nearest - identical to the reference except among exact
    duplicates of a row, whose order torch.topk leaves unspecified).  `normalized=False` (raw squared distances of unnormalised
    rows) has no HIP kernel - GMF never calls it - and raises."""
    if not normalized:
        raise NotImplementedError("gmf_amd.knn: normalized=False (distances of unnormalised rows) has no HIP kernel; GMF only calls "
                                  "knn(..., normalized=True) on unit features (PointDSC.py:229,327)")
    x = require_cuda_f32(x, "x").contiguous()
    bs, N, C = x.shape
    if C != 128:
        raise NotImplementedError("gmf_amd.knn: HIP kernel is built for 128-d features")
    kk = k if ignore_self else k - 1                   # neighbours behind rank 0
    out = torch.empty((bs, N, k), device=x.device, dtype=torch.int32)
    rows = torch.arange(N, device=x.device, dtype=torch.int32).repeat(bs, 1).contiguous()
    if ignore_self:
        nb = out
    else:
        out[:, :, 0] = rows
        nb = torch.empty((bs, N, max(kk, 1)), device=x.device, dtype=torch.int32)
    if kk > 0:
        h, st = handle_and_stream(x)
        h.call("gmf_knn_rows", x.data_ptr(), rows.data_ptr(), bs, N, N, kk, nb.data_ptr(), st)
        if not ignore_self:
            out[:, :, 1:] = nb[:, :, :kk]
    return out.long()
