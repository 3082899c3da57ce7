"""Feature blending container, same shape as the reference's ``forger/train/stitching.py:18-25``."""
from . import ops


class BlendedFeatures:
    """Canvas features that are blended into a block's output with a soft mask:
    ``alpha * features + (1 - alpha) * other`` (stitching.py:24-25), evaluated by nb_blend_f32."""

    def __init__(self, features, alpha):
        self.features = features
        self.alpha = alpha

    def blend(self, other_features):
        return ops.blend(self.features, self.alpha, other_features)
