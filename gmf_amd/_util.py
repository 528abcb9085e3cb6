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


def handle_and_stream(t: torch.Tensor, check: bool = False):
    """(library handle, current stream) of t's device.  check: raise now if an EARLIER call on this device flagged a
    non-finite result (a host read of the handle's status word - no synchronisation)."""
    idx = dev_index(t)
    st = torch.cuda.current_stream(idx).cuda_stream
    # [r5] set_handle_per_stream: a non-default stream has a handle of its own (never while that stream is capturing a graph)
    own = _lib._per_stream and st and not torch.cuda.is_current_stream_capturing()
    h = _lib.handle_for(idx, st if own else 0)
    if check:
        h.raise_if_flagged("forward")
    return h, st


class WeightWatcher:
    """Cheap change detector for the packed-weight cache.

    Walking named_parameters() costs milliseconds per call on a 600-tensor model; the tensor list is therefore
    captured once (and again whenever Module._apply / load_state_dict may have replaced tensors - the owner calls
    `invalidate()` from those hooks) and a call only sums the in-place version counters (~0.1 ms)."""

    def __init__(self, module: torch.nn.Module, skip_prefix: str = None):
        self.module, self.skip = module, skip_prefix
        self.tensors = None

    def invalidate(self):
        self.tensors = None

    def version(self) -> int:
        if self.tensors is None:
            named = list(self.module.named_parameters()) + list(self.module.named_buffers())
            self.tensors = [t for n, t in named if not (self.skip and n.startswith(self.skip))]
            self.base = hash(tuple(t.data_ptr() for t in self.tensors))
        v = self.base
        for t in self.tensors:
            v += t._version
        return v


def params_version(module: torch.nn.Module) -> int:
    """One-shot form (small modules)."""
    v = 0
    for p in list(module.parameters()) + list(module.buffers()):
        v = (v * 1000003 + p._version * 31 + p.data_ptr()) & 0xFFFFFFFFFFFF
    return v
