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

    Only the configuration GMF uses is implemented (ignore_self=True, normalized=True, C=128;
    PointDSC.py:327)."""
    if not (ignore_self and normalized):
        raise NotImplementedError("gmf_amd.knn: GMF only calls knn(ignore_self=True, normalized=True) (PointDSC.py:327)")
    x = require_cuda_f32(x, "x").contiguous()
    bs, N, C = x.shape
    if C != 128:
        raise NotImplementedError("gmf_amd.knn: HIP kernel is built for 128-d features")
    rows = torch.arange(N, device=x.device, dtype=torch.int32).repeat(bs, 1).contiguous()
    out = torch.empty((bs, N, k), device=x.device, dtype=torch.int32)
    h, st = handle_and_stream(x)
    h.call("gmf_knn_rows", x.data_ptr(), rows.data_ptr(), bs, N, N, k, out.data_ptr(), st)
    return out.long()
