"""Golden vectors for the augmentation pipeline (row f4): the REFERENCE ``training.augment.AugmentPipe`` on CPU (build
container only) in its deterministic ``debug_percentile`` mode (every random draw replaced by its value at that
percentile), for the 'bgc' configuration (blit + geometry + colour), with image-space filtering and cutout added, on RGB and
single-channel batches.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_augment.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import thirdparty.stylegan2_ada_pytorch  # noqa: E402,F401
from thirdparty.stylegan2_ada_pytorch.training import augment as ref_aug  # noqa: E402

BGC = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
CONFIGS = {"bgc": BGC, "bgcfc": dict(BGC, imgfilter=1, cutout=1), "color": dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
           "geom": dict(scale=1, rotate=1, aniso=1, xfrac=1), "filter": dict(imgfilter=1, imgfilter_bands=[1, 0, 1, 1])}


def main():
    rng = np.random.RandomState(77)
    out = {"img3": rng.randn(2, 3, 32, 32).astype(np.float32), "img1": rng.randn(2, 1, 32, 32).astype(np.float32)}
    for name, kw in CONFIGS.items():
        pipe = ref_aug.AugmentPipe(**kw)
        for pct in (0.15, 0.6, 0.85):
            for key in ("img3", "img1"):
                y = pipe(torch.tensor(out[key]), debug_percentile=pct)
                out[f"{name}_{key}_{int(pct * 100)}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("augment.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "augment.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
