"""MI355X-native (gfx950) implementation of the NeuBE stroke-generator forward path.

``config``, ``weights`` and ``synthetic`` are pure Python/numpy.  ``ops`` and ``networks`` drive the
hand-written HIP kernels in ``csrc/`` through the C ABI in ``include/neube_hip.h`` and raise if the
library is missing -- there is no CPU or torch fallback on the product path.
"""
from .config import GeneratorConfig, style1_config, tiny_config  # noqa: F401

__all__ = ["GeneratorConfig", "style1_config", "tiny_config"]
