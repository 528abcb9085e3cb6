"""PerceiverIO drop-in for the DGR plugin surface
(reference: GMF_DeepGlobalRegistration/*/model/perceiver_io.py:139-221).

Identical to FusionLayer except that ``to_out`` maps the head back to the QUERY width
(perceiver_io.py:83).  HIP kernels cover both instances DGR builds: ``image_fusion`` with
latent_dim = dim = 128, d_head = 64 (resunet_new.py:618-626) and the bottleneck ``perceiver_io`` with
latent_dim = 256, dim = 128, d_head = 128 (resunet_new.py:516-525).  Other widths raise
NotImplementedError - there is deliberately no fallback.
"""
from .fusion_layer import FusionLayer


class PerceiverIO(FusionLayer):
    _OUT_TO_QUERY = True
