"""Generator hyper-parameters for the NeuBE stroke generator ("style1 checkpoint shapes").

The reference builds its generator from ``train_flags.txt:1-20`` through
``thirdparty/stylegan2_ada_pytorch/train.py:327-353`` and
``training/networks_modified.py:43-118`` (SynthesisNetwork.__init__).  This module restates the
shape rules only (no arithmetic): which blocks exist, how many channels each layer has, where the
geometry features are concatenated and how many ``ws`` each block consumes.
"""
from __future__ import annotations

import dataclasses
import functools
import json
import math
from typing import List, Optional, Tuple


@dataclasses.dataclass(frozen=True)
class LayerSpec:
    """One modulated 3x3 convolution of the synthesis network (reference SynthesisLayer, networks.py:302-391)."""
    name: str            # e.g. "synthesis.b64.conv0"
    block_res: int       # resolution of the block (= output resolution of the layer)
    up: int              # 2 for conv0 (transposed conv + FIR), 1 for conv1
    in_channels: int     # including concatenated geometry channels
    out_channels: int
    geom_channels: int   # how many of in_channels come from the geometry feature (tail of the channel axis)
    w_index: int         # index into ws[:, w_index] that feeds this layer's affine

    @property
    def in_res(self) -> int:
        return self.block_res // self.up


@dataclasses.dataclass(frozen=True)
class GeneratorConfig:
    z_dim: int = 64
    c_dim: int = 0
    w_dim: int = 64
    img_resolution: int = 256
    img_channels: int = 3
    mapping_layers: int = 4              # cfg='auto' -> spec.map (train.py:266,342)
    mapping_lr_multiplier: float = 0.01  # networks.py:225
    channel_base: int = 16384            # fmaps=0.5 * 32768 (train.py:338)
    channel_max: int = 128               # train_flags.txt:16
    conv_clamp: Optional[float] = 256.0  # train.py:344
    geom_feature_channels: Tuple[int, ...] = (16, 256)      # sauto encoder, SURVEY 8
    geom_feature_resolutions: Tuple[int, ...] = ()          # default filled in __post_init__: (R/8, R/4)
    resample_filter: Tuple[int, ...] = (1, 3, 3, 1)

    def __post_init__(self):
        r = self.img_resolution
        assert r >= 4 and r & (r - 1) == 0, "img_resolution must be a power of two >= 4"
        if not self.geom_feature_resolutions and self.geom_feature_channels:
            object.__setattr__(self, "geom_feature_resolutions", (r // 8, r // 4))
        assert len(self.geom_feature_resolutions) == len(self.geom_feature_channels)
        assert self.c_dim == 0, "conditioning labels are not part of the NeuBE path (c_dim=0)"

    # ---- shape rules (networks_modified.py:63-118) ----
    @functools.cached_property
    def block_resolutions(self) -> List[int]:
        return [2 ** i for i in range(2, int(math.log2(self.img_resolution)) + 1)]

    def channels(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)

    def geom_channels_at(self, res: int) -> int:
        """Geometry channels concatenated after the block of resolution ``res`` (networks_modified.py:190-219)."""
        if res in self.geom_feature_resolutions:
            return self.geom_feature_channels[self.geom_feature_resolutions.index(res)]
        return 0

    @functools.cached_property
    def layers(self) -> List[LayerSpec]:
        """All modulated-conv layers in execution order (computed once: the forward pass consults it per layer)."""
        out: List[LayerSpec] = []
        w = 0
        for res in self.block_resolutions:
            oc = self.channels(res)
            if res > 4:
                g = self.geom_channels_at(res // 2)
                ic = self.channels(res // 2) + g
                out.append(LayerSpec(f"synthesis.b{res}.conv0", res, 2, ic, oc, g, w))
                w += 1
            out.append(LayerSpec(f"synthesis.b{res}.conv1", res, 1, oc, oc, 0, w))
            w += 1
        return out

    @property
    def torgb_w_index(self) -> int:
        return len(self.layers)

    @property
    def num_ws(self) -> int:
        return len(self.layers) + 1  # + the last block's torgb (networks_modified.py:115-117)

    @property
    def torgb_name(self) -> str:
        return f"synthesis.b{self.img_resolution}.torgb"

    # ---- work model (SURVEY 8 / BASELINE.md 2) ----
    def macs_per_patch(self) -> int:
        """Multiply-adds of one patch; up-layers counted at their non-zero transposed-conv MACs + 4x4 FIR."""
        total = 0
        for l in self.layers:
            if l.up == 2:
                total += l.in_res ** 2 * l.out_channels * l.in_channels * 9   # transposed conv, non-zero taps
                total += l.block_res ** 2 * l.out_channels * 16                # 4x4 FIR
            else:
                total += l.block_res ** 2 * l.out_channels * l.in_channels * 9
            total += l.in_channels * self.w_dim                               # affine
            total += 2 * l.out_channels * l.in_channels * 9                   # modulate + demodulate
        c = self.channels(self.img_resolution)
        total += self.img_resolution ** 2 * 3 * c + (c + 9) * self.w_dim
        total += self.mapping_layers * self.w_dim * self.w_dim
        return total

    def activation_bytes_per_patch(self) -> int:
        """Compulsory fp32 activation traffic if every layer reads its input and writes its output once."""
        total = 0
        for l in self.layers:
            total += 4 * (l.in_channels * l.in_res ** 2 + l.out_channels * l.block_res ** 2)
        r = self.img_resolution
        total += 4 * (self.channels(r) * r * r + 3 * r * r + 3 * r * r)
        return total

    def to_json(self) -> str:
        return json.dumps(dataclasses.asdict(self))

    @staticmethod
    def from_json(s: str) -> "GeneratorConfig":
        d = json.loads(s)
        for k in ("geom_feature_channels", "geom_feature_resolutions", "resample_filter"):
            d[k] = tuple(d[k])
        return GeneratorConfig(**d)


def style1_config(resolution: int = 256) -> GeneratorConfig:
    """The shipped hyper-parameters at a given output resolution (128 = as shipped, 256 = BASELINE metric)."""
    return GeneratorConfig(img_resolution=resolution)


def tiny_config(resolution: int = 32) -> GeneratorConfig:
    """A small net with the same structure (geometry injection at R/8 and R/4) for fast unit tests."""
    return GeneratorConfig(z_dim=32, w_dim=32, img_resolution=resolution, channel_base=resolution * 16,
                           channel_max=32, geom_feature_channels=(4, 8))
