"""PerceiverIO drop-in for the DGR plugin surface
(reference: GMF_DeepGlobalRegistration/*/model/perceiver_io.py:139-221).

Identical to FusionLayer except that ``to_out`` maps the head back to the QUERY width
(perceiver_io.py:83).  The HIP kernels currently cover the 128/128/64 configuration used by
``image_fusion`` (resunet_new.py:618-626); the 256-wide bottleneck instance (resunet_new.py:516-525)
raises NotImplementedError - there is deliberately no fallback.
"""
from .fusion_layer import FusionLayer


class PerceiverIO(FusionLayer):
    _OUT_TO_QUERY = True
