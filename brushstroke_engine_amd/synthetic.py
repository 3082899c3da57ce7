"""Seeded synthetic inputs for the generator path (SURVEY 8d "Synthetic inputs").

Used by ``bench.py``, the tests and the golden-vector generator, so that the same (cfg, n, seed)
yields the same tensors in the build container (where the reference is importable) and on the GPU
box (where it is not).  Everything is ``numpy.random.RandomState`` based.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .config import GeneratorConfig


def style_z(seed: int, z_dim: int) -> np.ndarray:
    """The reference's seed->z rule: ``np.random.RandomState(seed).randn(1, z_dim)`` (float64),
    ``forger/ui/brush.py:667-670``, ``forger/ui/library.py:222-225``."""
    return np.random.RandomState(seed=seed).randn(1, z_dim)


def batch_z(cfg: GeneratorConfig, n: int, first_seed: int = 0) -> np.ndarray:
    return np.concatenate([style_z(first_seed + i, cfg.z_dim) for i in range(n)], axis=0)


def stroke_masks(cfg: GeneratorConfig, n: int, seed: int = 0, density: float = 0.08) -> np.ndarray:
    """[n,1,R,R] float32 geometry guidance, 1 = background, 0 = stroke: Bernoulli(density) dots
    box-blurred 5x5 and thresholded."""
    rs = np.random.RandomState(seed)
    r = cfg.img_resolution
    dots = (rs.rand(n, r, r) < density / 4).astype(np.float32)
    pad = np.pad(dots, ((0, 0), (2, 2), (2, 2)))
    acc = np.zeros_like(dots)
    for dy in range(5):
        for dx in range(5):
            acc += pad[:, dy:dy + r, dx:dx + r]
    return (1.0 - (acc > 0.5).astype(np.float32))[:, None]


def geom_features(cfg: GeneratorConfig, n: int, seed: int = 0) -> List[np.ndarray]:
    """Stand-in for the geometry encoder's outputs (``forger/experimental/autoenc/base.py:123-134``):
    one [n, C_i, res_i, res_i] float32 map per injection point.  Values are post-LeakyReLU-like
    (mostly positive, O(1)); the encoder itself is row f1 of SURVEY 8f."""
    rs = np.random.RandomState(seed + 7919)
    out = []
    for c, res in zip(cfg.geom_feature_channels, cfg.geom_feature_resolutions):
        a = rs.randn(n, c, res, res).astype(np.float32)
        out.append(np.where(a > 0, a, 0.01 * a).astype(np.float32) * np.float32(0.7))
    return out


def positions(cfg: GeneratorConfig, n: int, seed: int = 0, limit: int = 4096) -> np.ndarray:
    """[n,2] int64 (y,x) patch positions, uniform in [0, limit) (exercises the noise shift, note C)."""
    rs = np.random.RandomState(seed + 104729)
    return rs.randint(0, limit, size=(n, 2)).astype(np.int64)
