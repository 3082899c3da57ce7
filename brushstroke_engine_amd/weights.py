"""Flat weight container for the NeuBE generator.

The reference persists networks as pickles that embed module *source*
(``torch_utils/persistence.py:118-126``), which cannot be read without the reference tree.  This
build uses a flat ``name -> float32 array`` dict whose names are exactly the reference's
``state_dict()`` keys ("Weights / state names"), stored as ``.npz`` with the
:class:`GeneratorConfig` JSON under ``__config__``.

``random_state_dict`` creates seeded synthetic weights (no checkpoints ship with the reference,
``.gitignore:47``).  It deliberately randomises the parameters the reference initialises to zero
(``noise_strength``, biases, ``color_bias``: networks.py:359-360, 403, 447-448) so that those code
paths are exercised (SURVEY 8a note D).
"""
from __future__ import annotations

import io
from typing import Dict

import numpy as np

from .config import GeneratorConfig

StateDict = Dict[str, np.ndarray]


def linspace01_f32(n: int) -> np.ndarray:
    """``torch.linspace(0, 1, n)`` in float32: ATen evaluates ``start + step*i`` for the lower half and
    ``end - step*(n-1-i)`` for the upper half with a fused multiply-add (one rounding), which the
    float64 product below reproduces bit for bit (checked against torch for n = 4..1024)."""
    if n == 1:
        return np.zeros([1], np.float32)
    step = np.float64(np.float32(1.0) / np.float32(n - 1))
    i = np.arange(n)
    lo = (step * i).astype(np.float32)
    hi = (1.0 - step * (n - 1 - i)).astype(np.float32)
    return np.where(i < n // 2, lo, hi).astype(np.float32)


def make_noise_grid(res: int) -> np.ndarray:
    """Reference ``create_sampling_grid`` (networks.py:295-299): meshgrid(ij) stacked as (x_i, y_j)."""
    lin = linspace01_f32(res)
    xv, yv = np.meshgrid(lin, lin, indexing="ij")
    return np.stack([xv, yv], axis=-1)[None].astype(np.float32)


def make_resample_filter(taps=(1, 3, 3, 1)) -> np.ndarray:
    """Reference ``upfirdn2d.setup_filter`` (upfirdn2d.py:72-116) for a short 1-D tap list: outer product / sum."""
    f = np.asarray(taps, np.float32)
    f2 = np.outer(f, f).astype(np.float32)
    return (f2 / f2.sum(dtype=np.float32)).astype(np.float32)


def random_state_dict(cfg: GeneratorConfig, seed: int = 0) -> StateDict:
    """Seeded synthetic weights with the reference's names and shapes.

    Uses ``numpy.random.RandomState`` (bit-stable across numpy versions) and draws in a fixed order,
    so the same (cfg, seed) gives the same network here, in the golden-vector generator (which loads
    this dict into the reference ``Generator`` with ``strict=True``) and on the GPU box.
    """
    rs = np.random.RandomState(seed)
    sd: StateDict = {}

    def randn(*shape, scale=1.0):
        return (rs.randn(*shape) * scale).astype(np.float32)

    lr = cfg.mapping_lr_multiplier
    for i in range(cfg.mapping_layers):
        sd[f"mapping.fc{i}.weight"] = randn(cfg.w_dim, cfg.z_dim if i == 0 else cfg.w_dim, scale=1.0 / lr)
        sd[f"mapping.fc{i}.bias"] = randn(cfg.w_dim, scale=0.1 / lr)
    sd["mapping.w_avg"] = randn(cfg.w_dim, scale=0.1)

    filt = make_resample_filter(cfg.resample_filter)
    c4 = cfg.channels(4)
    sd["synthesis.b4.const"] = randn(c4, 4, 4)
    for res in cfg.block_resolutions:
        sd[f"synthesis.b{res}.resample_filter"] = filt.copy()
    for l in cfg.layers:
        sd[f"{l.name}.weight"] = randn(l.out_channels, l.in_channels, 3, 3)
        sd[f"{l.name}.noise_strength"] = np.float32(rs.uniform(0.02, 0.1)).reshape(())
        sd[f"{l.name}.bias"] = randn(l.out_channels, scale=0.1)
        sd[f"{l.name}.noise_grid"] = make_noise_grid(l.block_res)
        sd[f"{l.name}.resample_filter"] = filt.copy()
        sd[f"{l.name}.noise_const"] = randn(l.block_res, l.block_res)
        sd[f"{l.name}.affine.weight"] = randn(l.in_channels, cfg.w_dim)
        sd[f"{l.name}.affine.bias"] = (1.0 + randn(l.in_channels, scale=0.1)).astype(np.float32)
    t = cfg.torgb_name
    c = cfg.channels(cfg.img_resolution)
    sd[f"{t}.weight"] = randn(3, c, 1, 1, scale=3.0)   # wider logits: softmax far from uniform
    sd[f"{t}.bias"] = randn(3, scale=0.5)
    sd[f"{t}.color_bias"] = randn(9, scale=0.5)
    sd[f"{t}.affine.weight"] = randn(c + 9, cfg.w_dim)
    sd[f"{t}.affine.bias"] = (1.0 + randn(c + 9, scale=0.1)).astype(np.float32)
    return sd


def hdr_state_dict(cfg: GeneratorConfig, seed: int = 0, act_scale: float = 60.0, logit_scale: float = 0.25) -> StateDict:
    """High-dynamic-range variant of :func:`random_state_dict` for precision tests: the learned constant, the biases and
    the noise strengths are scaled so that activations run at an rms of ~50 and several layers reach ``conv_clamp`` = 256
    (trained StyleGAN2 generators do -- that is why the reference clamps, networks.py:389), and the ToRGB weights so that
    the triad logits span about +-20 (saturated and steep softmax regions both present).  The split-f16 / fp8 conv
    modes have RELATIVE error bounds, so this is where their absolute pixel error is largest."""
    sd = random_state_dict(cfg, seed)
    a = np.float32(act_scale)
    sd["synthesis.b4.const"] = sd["synthesis.b4.const"] * a
    for l in cfg.layers:
        sd[f"{l.name}.bias"] = (sd[f"{l.name}.bias"] * (a * 3)).astype(np.float32)
        sd[f"{l.name}.noise_strength"] = (sd[f"{l.name}.noise_strength"] * (a * 3)).astype(np.float32)
    t = cfg.torgb_name
    sd[f"{t}.weight"] = (sd[f"{t}.weight"] * np.float32(logit_scale / 3.0 / act_scale * 20)).astype(np.float32)
    return sd


def trained_like_state_dict(cfg: GeneratorConfig, seed: int = 0) -> StateDict:
    """Variant of :func:`random_state_dict` with the STATISTICS of a trained StyleGAN2 generator instead of i.i.d. Gaussian
    weights (there is no checkpoint in the reference tree): every conv weight gets log-normal per-input-channel and
    per-output-channel scales (sigma = 1: a few channels carry most of the energy, heavy tails), two input channels per
    layer are DOMINANT styles (affine bias x8, so the demodulated contraction is dominated by a handful of terms and the
    errors of the split products do not average out over 128+ channels), the affine weights are 3x larger (styles vary
    strongly with w) and the noise strengths 5x.  Used by the f8 margin fixtures (tests/golden/gen_trained_r128.npz)."""
    sd = random_state_dict(cfg, seed)
    rs = np.random.RandomState(seed + 7919)
    for l in cfg.layers:
        w = sd[f"{l.name}.weight"]
        ci = np.exp(rs.randn(l.in_channels)).astype(np.float32)
        co = np.exp(rs.randn(l.out_channels)).astype(np.float32)
        sd[f"{l.name}.weight"] = (w * co[:, None, None, None] * ci[None, :, None, None]).astype(np.float32)
        ab = sd[f"{l.name}.affine.bias"].copy()
        dom = rs.choice(l.in_channels, size=min(2, l.in_channels), replace=False)
        ab[dom] *= np.float32(8.0)
        sd[f"{l.name}.affine.bias"] = ab
        sd[f"{l.name}.affine.weight"] = (sd[f"{l.name}.affine.weight"] * np.float32(3.0)).astype(np.float32)
        sd[f"{l.name}.noise_strength"] = (sd[f"{l.name}.noise_strength"] * np.float32(5.0)).astype(np.float32)
    return sd


def expected_shapes(cfg: GeneratorConfig) -> Dict[str, tuple]:
    return {k: tuple(v.shape) for k, v in random_state_dict(cfg, 0).items()}


def validate_state_dict(cfg: GeneratorConfig, sd: StateDict) -> None:
    exp = expected_shapes(cfg)
    missing = sorted(set(exp) - set(sd))
    if missing:
        raise KeyError(f"weight container is missing {len(missing)} tensors, e.g. {missing[:4]}")
    for k, shp in exp.items():
        if tuple(np.shape(sd[k])) != shp:
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(np.shape(sd[k]))}")


def save_weights(path, cfg: GeneratorConfig, sd: StateDict) -> None:
    validate_state_dict(cfg, sd)
    arrays = {k: np.asarray(v, np.float32) for k, v in sd.items()}
    arrays["__config__"] = np.frombuffer(cfg.to_json().encode("utf-8"), dtype=np.uint8)
    np.savez(path, **arrays)


def load_weights(path):
    with np.load(path) as z:
        cfg = GeneratorConfig.from_json(bytes(z["__config__"]).decode("utf-8"))
        sd = {k: np.array(z[k], np.float32) for k in z.files if k != "__config__"}
    validate_state_dict(cfg, sd)
    return cfg, sd


def weights_to_bytes(cfg: GeneratorConfig, sd: StateDict) -> bytes:
    buf = io.BytesIO()
    save_weights(buf, cfg, sd)
    return buf.getvalue()
