"""ctypes binding of libgmf_hip.so (C ABI: include/gmf_hip.h).

There is NO fallback: if the shared library is missing, or a call returns a non-zero status,
a RuntimeError is raised (the reference's assert/exception semantics).
A non-finite result (an activation outside the fp16 range of the split MFMA operands) sets a sticky status bit on the
device: the next forward of a drop-in module on that device, or `gmf_amd.check_status()`, raises.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgmf_hip.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_vp = C.c_void_p
_ll = C.c_longlong


class EncoderWeights(C.Structure):
    """Mirror of `struct gmf_encoder_weights` (include/gmf_hip.h)."""
    _fields_ = [
        ("num_layers", C.c_int),
        ("f1_ctx_wst", _vp), ("f1_ctx_vec", _vp),
        ("f1_attn_wst", _vp), ("f1_attn_vec", _vp),
        ("f1_ff_wst", _vp), ("f1_ff_vec", _vp),
        ("ctx_wst", _vp), ("ctx_vec", _vp), ("ctx_wst_stride", C.c_int), ("ctx_vec_stride", C.c_int),
        ("attn_wst", _vp), ("attn_vec", _vp), ("attn_wst_stride", C.c_int), ("attn_vec_stride", C.c_int),
        ("ff_wst", _vp), ("ff_vec", _vp), ("ff_wst_stride", C.c_int), ("ff_vec_stride", C.c_int),
        ("front_wst", _vp), ("front_vec", _vp), ("front_wst_stride", C.c_int), ("front_vec_stride", C.c_int),
        ("tail_wst", _vp), ("tail_vec", _vp), ("tail_wst_stride", C.c_int), ("tail_vec_stride", C.c_int),
        ("head_wst", _vp), ("head_vec", _vp),
        ("sigma_d", C.c_float),
        ("front_wst_h2", _vp), ("ctx_wst_h2", _vp), ("attn_wst_h2", _vp), ("ff_wst_h2", _vp),
        ("f1_ctx_wst_h2", _vp), ("f1_attn_wst_h2", _vp), ("f1_ff_wst_h2", _vp),
        ("tail_wst_h2", _vp),
        ("pv_guard", _vp),
    ]


class Tensor(C.Structure):
    """Mirror of `struct gmf_tensor` (include/gmf_hip.h): one named fp32 state_dict tensor."""
    _fields_ = [("name", C.c_char_p), ("data", _vp), ("ndim", C.c_int), ("shape", _ll * 4)]


class FusionWeights(C.Structure):
    """Mirror of `struct gmf_fusion_weights` (include/gmf_hip.h)."""
    _fields_ = [("latent_dim", C.c_int), ("d_head", C.c_int), ("pe", C.c_int), ("split_fp16", C.c_int),
                ("max_abs_scaled", C.c_float),
                ("ctx_wst", _vp), ("ctx_vec", _vp), ("attn_wst", _vp), ("attn_vec", _vp), ("ff_wst", _vp), ("ff_vec", _vp),
                ("ctx_wst_h2", _vp), ("attn_wst_h2", _vp), ("ff_wst_h2", _vp)]


GMF_PACK_DEVICE_TENSORS = 1
GMF_PACK_STANDALONE_BLOCK = 2
GMF_PACK_HOST_BLOCK = 4


def tensor_list(sd):
    """state_dict -> (ctypes array of gmf_tensor, keep-alive list).  Floating tensors only, as contiguous fp32 on the host."""
    import torch
    keep, items = [], []
    # [r4] tensors that live on a device travel to the host as ONE flat copy (a state_dict of the encoder is ~590 tensors: one
    # device-to-host copy and one synchronisation instead of 590 - ADVICE r3)
    host = {}
    dev = [(k, v) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and v.device.type != "cpu" and v.dim() <= 4]
    if dev:
        flat = torch.cat([v.detach().reshape(-1).to(torch.float32) for _, v in dev]).cpu()
        o = 0
        for k, v in dev:
            host[k] = flat[o:o + v.numel()].reshape(v.shape)
            o += v.numel()
    for k, v in sd.items():
        if not (torch.is_tensor(v) and v.is_floating_point()):
            continue
        t = host[k] if k in host else v.detach().to("cpu", torch.float32).contiguous()
        if t.dim() > 4:
            continue
        name = k.encode()
        keep += [t, name]
        e = Tensor()
        e.name, e.data, e.ndim = name, t.data_ptr(), t.dim()
        for d in range(4):
            e.shape[d] = t.shape[d] if d < t.dim() else 1
        items.append(e)
    arr = (Tensor * len(items))(*items)
    return arr, keep


class PoseParams(C.Structure):
    """Mirror of `struct gmf_pose_params` (include/gmf_hip.h)."""
    _fields_ = [
        ("num_seeds", C.c_int), ("k", C.c_int), ("num_iterations", C.c_int), ("use_nms", C.c_int),
        ("refine_iters", C.c_int),
        ("sigma", C.c_float), ("sigma_d", C.c_float), ("inlier_threshold", C.c_float),
        ("nms_radius", C.c_float), ("refine_threshold", C.c_float),
    ]


# name -> (restype, argtypes).  Every symbol declared in include/gmf_hip.h appears here.
SIGNATURES = {
    "gmf_abi_version": (C.c_int, []),
    "gmf_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gmf_destroy": (None, [_vp]),
    "gmf_last_error_string": (C.c_char_p, [_vp]),
    "gmf_workspace_bytes": (_ll, [_vp]),
    "gmf_set_workspace": (C.c_int, [_vp, _vp, _ll]),
    "gmf_workspace_wanted": (_ll, [_vp]),
    "gmf_status_read": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int]),
    "gmf_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "gmf_get_tuning": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "gmf_set_sigma_device": (C.c_int, [_vp, _vp]),
    "gmf_profile_enable": (C.c_int, [_vp, C.c_int]),
    "gmf_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gmf_pack_rows_p32": (C.c_int, [_vp, _vp, _ll, _ll, _ll, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_unpack_rows_p32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _ll, _ll, _ll, _vp]),
    "gmf_pack_pts8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "gmf_front_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gmf_scattn_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "gmf_scattn_forward_dense": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gmf_fusion_ctx_prepare": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "gmf_fusion_attn_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gmf_fusion_ff_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gmf_classifier_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gmf_encoder_pack_weights": (C.c_int, [_vp, C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gmf_packed_encoder_weights": (C.POINTER(EncoderWeights), [_vp]),
    "gmf_packed_encoder_info": (C.c_int, [_vp, _f32p, _f32p, _i32p, _f32p]),
    "gmf_packed_encoder_free": (None, [_vp]),
    "gmf_packed_encoder_bytes": (C.c_longlong, [_vp]),
    "gmf_packed_encoder_place": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _vp]),
    "gmf_fusion_pack_weights": (C.c_int, [_vp, C.POINTER(Tensor), C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gmf_packed_fusion_weights": (C.POINTER(FusionWeights), [_vp]),
    "gmf_packed_fusion_free": (None, [_vp]),
    "gmf_packed_fusion_bytes": (C.c_longlong, [_vp]),
    "gmf_packed_fusion_place": (C.c_int, [_vp, _vp, _vp, C.c_longlong, _vp]),
    "gmf_encoder_forward": (C.c_int, [_vp, C.POINTER(EncoderWeights), _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                      _vp, _vp, _vp, _vp]),
    "gmf_encoder_forward_ragged": (C.c_int, [_vp, C.POINTER(EncoderWeights), _vp, _vp, _vp, _vp, _vp, _i32p, C.c_int, C.c_int,
                                             _vp, _vp, _vp, _vp]),
    "gmf_pose_head_ragged": (C.c_int, [_vp, C.POINTER(PoseParams), C.c_double, _vp, _vp, _vp, _vp, _i32p, C.c_int, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp]),
    "gmf_nonlocal_block_forward": (C.c_int, [_vp, C.POINTER(EncoderWeights), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                             C.c_int, C.c_int, C.c_int, _vp]),
    "gmf_fusion_layer_forward": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll,
                                           _ll, _vp, _ll, _ll, _ll, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "gmf_pose_head": (C.c_int, [_vp, C.POINTER(PoseParams), _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
    "gmf_pick_seeds": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _vp, _vp]),
    "gmf_knn_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_knn_from_distances": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_nn_match": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "gmf_procrustes_batched": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    "gmf_post_refinement": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp]),
    "gmf_weighted_procrustes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_float, _vp, _vp, _vp]),
    "gmf_global_registration": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                          C.c_double, _vp, _vp, _vp, C.c_int, _vp]),
    "gmf_similarity_matrix": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp]),
    "gmf_spectral_matching_loss": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_spectral_matching_loss_fused": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp]),
    "gmf_spectral_matching_backward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp, _vp]),
    "gmf_classification_loss": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_stem_forward": (C.c_int, [_vp, _vp, _ll, _ll, _ll, _ll, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gmf_gemm_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int] + [_ll] * 6 + [C.c_int, C.c_float, C.c_int, _vp]),
    "gmf_lcpe": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gmf_layernorm_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, C.c_int, _vp]),
    "gmf_layernorm_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, C.c_int, _vp]),
    "gmf_softmax_rows": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _ll, C.c_int, C.c_float, _vp]),
    "gmf_geglu": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _ll, C.c_int, _vp]),
    "gmf_colsum": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _ll, C.c_int, _vp, C.c_int, _vp, _vp]),
    "gmf_batchnorm_train_forward": (C.c_int, [_vp] * 9 + [_ll, C.c_int, C.c_float, C.c_float, C.c_int, _vp]),
    "gmf_batchnorm_train_backward": (C.c_int, [_vp] * 10 + [_ll, C.c_int, _vp]),
    "gmf_normalize_rows": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _ll, C.c_int, _vp]),
    "gmf_relu_backward": (C.c_int, [_vp, _vp, _vp, _vp, _ll, _vp]),
    "gmf_classification_backward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_spectral_matching_dense_backward": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gmf_similarity_backward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp]),
    "gmf_conv_nhwc": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp]),
    "gmf_bias_relu_nhwc": (C.c_int, [_vp, _vp, _vp, _vp, C.c_longlong, C.c_int, _vp]),
    "gmf_transformation_loss": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    "gmf_compat_dense": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    "gmf_transformation_loss_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "gmf_pose_head_backward": (C.c_int, [_vp, C.POINTER(PoseParams), _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "gmf_weighted_procrustes_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_float, _vp, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


def load_library() -> C.CDLL:
    """dlopen libgmf_hip.so and bind every entry point.  Raises RuntimeError if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"gmf_amd: {LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gmf_amd/csrc`.  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.gmf_abi_version() != 5:
            raise RuntimeError("gmf_amd: libgmf_hip.so ABI version mismatch")
        _lib = lib
        return lib


GMF_ERR_WORKSPACE = -6
GMF_STATUS_NONFINITE = 1
GMF_STATUS_PV_GUARDED = 2      # informational: the "pv_fp8" guard sent a (pair, layer) to the three-product form


class Handle:
    """Per-device library handle (`gmf_create`).  One per device per process; calls on it are serialised by the library.

    The library's workspace is a torch tensor (`gmf_set_workspace`): the memory belongs to torch's caching allocator, not
    to a private hipMalloc block.  It grows on demand - a call that needs more returns GMF_ERR_WORKSPACE, `call` allocates
    what `gmf_workspace_wanted` says and repeats the call once."""

    def __init__(self, device: int):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.gmf_create(int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"gmf_amd: gmf_create(device={device}) failed with status {rc} "
                               "(no HIP device?) - the HIP path is mandatory, there is no CPU fallback")
        self.h = h
        self.device = device
        self._ws = None               # torch.uint8 tensor behind gmf_set_workspace
        self.torch_workspace = True   # False: the library keeps its own hipMalloc block (non-torch hosts do the same)

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.gmf_last_error_string(self.h)
            raise RuntimeError(f"gmf_amd: {what} failed (status {rc}): {msg.decode() if msg else ''}")

    def _grow_workspace(self):
        import torch
        want = int(self.lib.gmf_workspace_wanted(self.h))
        # release the old block first (torch's allocator is stream-ordered: work already queued on the stream keeps it valid)
        self.check(self.lib.gmf_set_workspace(self.h, None, 0), "gmf_set_workspace")
        if self._ws is not None:
            self._ws.record_stream(torch.cuda.current_stream(self.device))
        self._ws = None
        self._ws = torch.empty(want, dtype=torch.uint8, device=torch.device("cuda", self.device))
        self.check(self.lib.gmf_set_workspace(self.h, self._ws.data_ptr(), want), "gmf_set_workspace")

    def call(self, name: str, *args):
        fn = getattr(self.lib, name)
        if self.torch_workspace and self._ws is None:
            # start with an empty caller-provided workspace, so the library never allocates a block of its own
            import torch
            self._ws = torch.empty(256, dtype=torch.uint8, device=torch.device("cuda", self.device))
            self.check(self.lib.gmf_set_workspace(self.h, self._ws.data_ptr(), 256), "gmf_set_workspace")
        rc = fn(self.h, *args)
        if rc == GMF_ERR_WORKSPACE and self.torch_workspace:
            self._grow_workspace()
            rc = fn(self.h, *args)
        self.check(rc, name)

    def status(self, clear: bool = False) -> int:
        """The handle's sticky status word (no device synchronisation: it reflects the work that has FINISHED)."""
        f = C.c_int(0)
        self.check(self.lib.gmf_status_read(self.h, C.byref(f), 1 if clear else 0), "gmf_status_read")
        return f.value

    def raise_if_flagged(self, where: str):
        f = self.status(clear=False)
        if f & GMF_STATUS_NONFINITE:
            self.status(clear=True)
            raise RuntimeError(
                f"gmf_amd: {where}: a non-finite value (NaN / inf) reached an output of an earlier call on this device - "
                "an input was non-finite, or an activation left the range of the split-fp16 MFMA operands (|x| < 65504; "
                "INTEGRATION.md, 'Supported value range').  The outputs of that call are not valid.")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.gmf_destroy(self.h)
                self.h = None
        except Exception:
            pass


def check_status(device=None, synchronize: bool = True):
    """Raise if any call on `device` since the last check produced a non-finite logit, feature norm or fusion output.
    Synchronises the device first (pass synchronize=False after a synchronisation of your own)."""
    import torch
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
    with _handles_lock:
        extra = [hd for (d, _), hd in _stream_handles.items() if d == idx]
    if idx not in _handles and not extra:
        return
    if synchronize:
        torch.cuda.synchronize(idx)
    for hd in ([_handles[idx]] if idx in _handles else []) + extra:
        hd.raise_if_flagged("check_status")


_handles: Dict[int, Handle] = {}
# [r5] opt-in: one handle per (device, non-default stream) - see set_handle_per_stream
_stream_handles: "collections.OrderedDict" = collections.OrderedDict()
_per_stream = False
_MAX_STREAM_HANDLES = 8
_handles_lock = threading.Lock()


def set_handle_per_stream(on: bool = True):
    """Serving: let forwards issued on DIFFERENT torch streams overlap on the device.  A library handle serialises its calls (one
    workspace), and by default a process has one handle per device - two B = 1 forwards on two streams then run one after the other
    although each fills a tenth of the chip.  With this switch on, every non-default stream gets a handle (and workspace) of its
    own, created at its first call and kept for the 8 most recently used streams; the default stream keeps the device's base handle.
    Two streams: 1.85 x the forwards per second at N = 1000, three: 2.3 x (tools/concurrent_streams.py).  Results do not depend on it.
    Not for processes that capture HIP graphs (a capture stream would meet a handle whose workspace was never sized): calls made
    while a stream is capturing always take the base handle.  `gmf_set_tuning` knobs belong to a handle: set through
    `handle_for(device)` they apply to the default stream's."""
    global _per_stream
    _per_stream = bool(on)
    if not on:
        with _handles_lock:
            _stream_handles.clear()


def handle_for(device_index: Optional[int], stream_ptr: int = 0) -> Handle:
    idx = 0 if device_index is None else int(device_index)
    if _per_stream and stream_ptr:
        key = (idx, int(stream_ptr))
        with _handles_lock:
            hd = _stream_handles.get(key)
            if hd is None:
                hd = Handle(idx)
                _stream_handles[key] = hd
                while len(_stream_handles) > _MAX_STREAM_HANDLES:
                    _stream_handles.popitem(last=False)         # least recently used: its workspace goes back to torch's allocator
            else:
                _stream_handles.move_to_end(key)
        return hd
    hd = _handles.get(idx)
    if hd is None:
        hd = Handle(idx)
        _handles[idx] = hd
    return hd


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
