"""gmf_amd - MI355X-native GMF multimodal-fusion hot path (PointDSC / DGR plugin surface).

Python host modules with the reference's constructor signatures, ``forward`` signatures and
``state_dict`` keys; all compute goes through libgmf_hip.so (hand-written HIP kernels for gfx950)
over the C ABI declared in include/gmf_hip.h.  There is no CPU or eager-PyTorch fallback.
"""
from .fusion_layer import FusionLayer            # noqa: F401
from .perceiver_io import PerceiverIO            # noqa: F401
from .pointdsc import NonLocalBlock, NonLocalNet, PointDSC, ImageEncoder   # noqa: F401
from .common import rigid_transform_3d, knn      # noqa: F401
from .registration import (weighted_procrustes, weighted_procrustes_batched, GlobalRegistration,  # noqa: F401
                           global_registration_batched, argmin_se3_squared_dist, Transformation, ortho2rotation)
from .matching import nn_match, find_knn_gpu    # noqa: F401
from . import se3 as SE3                         # noqa: F401
from ._lib import check_status, set_handle_per_stream   # noqa: F401
from .losses import ClassificationLoss, SpectralMatchingLoss, TransformationLoss, similarity_matrix   # noqa: F401

__all__ = ["FusionLayer", "PerceiverIO", "NonLocalBlock", "NonLocalNet", "PointDSC", "ImageEncoder",
           "rigid_transform_3d", "knn", "weighted_procrustes", "weighted_procrustes_batched", "GlobalRegistration", "global_registration_batched", "argmin_se3_squared_dist", "Transformation", "ortho2rotation", "nn_match",
           "find_knn_gpu", "SE3", "ClassificationLoss", "SpectralMatchingLoss", "TransformationLoss",
           "similarity_matrix", "check_status", "set_handle_per_stream"]
