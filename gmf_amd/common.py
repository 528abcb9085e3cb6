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


def _knn_general(x, kk, normalized):
    """The kk nearest rows BEHIND rank 0 for any feature width and either distance form (common.py:64-69): the distance rows
    xx_j - 2 x_i.x_j (+ xx_i, constant along a row: it does not move the order and is left out) from the library's fp32-MFMA GEMM -
    exact fp32 products, no operand-range assumptions - then the library's row selection (gmf_knn_from_distances)."""
    from . import train as T
    bs, N, C = x.shape
    ld = ((N + 31) // 32) * 32
    nb = torch.empty((bs, N, kk), device=x.device, dtype=torch.int32)
    # the column term of the distances: xx_j (common.py:67), or the constant 2 of `2 - inner` for unit rows (:65)
    col = torch.full((bs, N), 2.0, device=x.device) if normalized else (x * x).sum(dim=-1)
    h, st = handle_and_stream(x)
    slab = max(1, min(bs, (1 << 30) // (N * ld)))               # at most 4 GiB of distance rows at a time
    dist = torch.empty((slab, N, ld), device=x.device, dtype=torch.float32)
    for b0 in range(0, bs, slab):
        nbat = min(slab, bs - b0)
        for b in range(nbat):
            T.gemm(x[b0 + b], x[b0 + b], tb=True, alpha=-2.0, bias=col[b0 + b], out=dist[b], m=N, n=N, k=C, lda=C, ldb=C, ldc=ld)
        h.call("gmf_knn_from_distances", dist.data_ptr(), nbat, N, N, kk, nb[b0:b0 + nbat].data_ptr(), st)
    return nb


def knn(x, k, ignore_self=False, normalized=True):
    """Indices of the k nearest rows (common.py:53-75): x [bs,N,C] -> [bs,N,k] int64.

    `normalized=True` (unit rows, distance 2 - 2 x x^T - what GMF feeds, PointDSC.py:229,327) at 128 channels runs the pose head's
    path (split-fp16 MFMA distance rows + threshold selection).  [r5] `normalized=False` (xx - 2 x x^T + xx^T, common.py:66-68: the
    form the reference's unused EdgeConv asks for, :94) and any other channel count form the distance rows with the fp32-MFMA GEMM
    instead (`_knn_general`).  `ignore_self=True`: top-(k+1) with rank 0 dropped; `ignore_self=False`: the row's own index (its
    distance is the row minimum: 0) followed by its k - 1 nearest - the reference's top-k except among exact duplicates of a row,
    whose order torch.topk leaves unspecified."""
    x = require_cuda_f32(x, "x").contiguous()
    bs, N, C = x.shape
    kk = k if ignore_self else k - 1                   # neighbours behind rank 0
    if kk > N - 1:
        raise RuntimeError(f"gmf_amd.knn: k = {k} needs more than {N} rows")
    general = (not normalized) or C != 128
    if general and kk > 63:
        raise NotImplementedError("gmf_amd.knn: more than 63 neighbours behind rank 0 are only built for unit 128-d rows")
    out = torch.empty((bs, N, k), device=x.device, dtype=torch.int32)
    rows = torch.arange(N, device=x.device, dtype=torch.int32).repeat(bs, 1).contiguous()
    if ignore_self:
        nb = out
    else:
        out[:, :, 0] = rows
        nb = torch.empty((bs, N, max(kk, 1)), device=x.device, dtype=torch.int32)
    if kk > 0:
        if general:
            got = _knn_general(x, kk, normalized)
            if ignore_self:
                out = got
            else:
                out[:, :, 1:] = got
        else:
            h, st = handle_and_stream(x)
            h.call("gmf_knn_rows", x.data_ptr(), rows.data_ptr(), bs, N, N, kk, nb.data_ptr(), st)
            if not ignore_self:
                out[:, :, 1:] = nb[:, :, :kk]
    return out.long()
