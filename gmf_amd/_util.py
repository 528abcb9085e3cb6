"""Shared host-side helpers for the drop-in modules."""
from __future__ import annotations

import torch

from . import _lib


def require_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"gmf_amd: `{name}` must live on a HIP device (got {t.device}); "
                           "the HIP path is mandatory, there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gmf_amd: `{name}` must be float32 (got {t.dtype})")
    return t


def dev_index(t: torch.Tensor) -> int:
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def handle_and_stream(t: torch.Tensor):
    idx = dev_index(t)
    return _lib.handle_for(idx), torch.cuda.current_stream(idx).cuda_stream


def params_version(module: torch.nn.Module) -> int:
    """Cheap change detector for the packed-weight cache: sum of tensor versions + storage pointers."""
    v = 0
    for p in list(module.parameters()) + list(module.buffers()):
        v = (v * 1000003 + p._version * 31 + p.data_ptr()) & 0xFFFFFFFFFFFF
    return v
