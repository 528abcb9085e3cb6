"""Descriptor-space nearest-neighbour matching: the step that produces the putative
correspondences on both plugin surfaces.

  nn_match(Fs, Ft)            PointDSC: datasets/ThreeDMatch.py:164-166, demo_registration.py:101-103
  find_knn_gpu(F0, F1, ...)   DGR: core/knn.py:23-74 (knn = 1), distances as core/metrics.py:62-69
"""
from __future__ import annotations

import torch

from ._util import handle_and_stream, require_cuda_f32


def _match(F0, F1, mode):
    F0 = require_cuda_f32(F0, "F0").contiguous()
    F1 = require_cuda_f32(F1, "F1").contiguous()
    if F0.dim() != 2 or F1.dim() != 2 or F0.shape[1] != F1.shape[1]:
        raise RuntimeError(f"gmf_amd.matching: expected [N0,d] and [N1,d], got {tuple(F0.shape)} / {tuple(F1.shape)}")
    if F0.shape[1] > 128:
        raise NotImplementedError("gmf_amd.matching: descriptor width above 128 has no HIP kernel")
    idx = torch.empty(F0.shape[0], device=F0.device, dtype=torch.int32)
    dist = torch.empty(F0.shape[0], device=F0.device, dtype=torch.float32)
    h, st = handle_and_stream(F0)
    h.call("gmf_nn_match", F0.data_ptr(), F1.data_ptr(), F0.shape[0], F1.shape[0], F0.shape[1], mode,
           idx.data_ptr(), dist.data_ptr(), st)
    return idx.long(), dist


def nn_match(src_desc, tgt_desc):
    """PointDSC matching for unit descriptors: (source_idx [Ns], source_dis [Ns]) with
    distance = sqrt(2 - 2 <s,t> + 1e-6) and argmin over the target rows."""
    return _match(src_desc, tgt_desc, 0)


def find_knn_gpu(F0, F1, nn_max_n=-1, knn=1, return_distance=False):
    """DGR find_knn_gpu.  `nn_max_n` only selects the reference's distance convention (L2 when chunked, squared L2
    otherwise); no chunking is needed here because the N0 x N1 matrix is never materialised."""
    if knn != 1:
        raise NotImplementedError("gmf_amd.find_knn_gpu: GMF-DGR only uses knn = 1 (deep_global_registration.py:300)")
    idx, dist = _match(F0, F1, 1 if nn_max_n > 1 else 2)
    if nn_max_n > 1:
        idx = idx[:, None]          # the reference concatenates [rows, knn] blocks (knn.py:40-41,62-63)
    return (idx, dist[:, None]) if return_distance else idx
