"""The forger loss items and the random stitcher of the training step: what the reference's
``Ggeom`` / ``Ggeom-warm`` phases and ``accumulate_gradients_stitch`` evaluate on the generator's debug dict
(``thirdparty/stylegan2_ada_pytorch/training/loss_modified.py:108-138, 181-203``).

* :class:`ForgerLosses` -- a weighted sum configured by the reference's strings, e.g. the shipped
  ``--geom_phase_losses='1.0*iou_inv(uvs)'`` / ``--geom_warmstart_losses='1.0*iou_inv(uvs)+1.0*iou(u)'``
  (``train_flags.txt:10-11``): ``forger/train/losses.py:37-233`` (container, string grammar) and the items ``iou``,
  ``iou_inv``, ``dice``, ``dice_inv``, ``l1``, ``gan``, ``rgb`` (``:341-377, 453-546, 634-666``).  The perceptual items
  (``lpips``, ``plpips``) need the LPIPS network, which neither the reference tree nor this build carries: they raise.
* :class:`CropHelper`, :class:`RandomStitcher` -- ``forger/train/stitching.py:28-267``: two overlapping crops of one
  drawing are generated with consistent noise positions and composited into each other.

Everything here is arithmetic on tensors the differentiable generator (:mod:`training`) returns -- torch ops, as in the
reference; the kernels sit below, in the generator and discriminator passes these phases run.
"""
from __future__ import annotations

import random
import re
from typing import Dict, List, Tuple

import torch

_COMPONENTS = {"canvas", "uvs", "u", "alpha", "fake_img", "color_0", "color_1", "color_2", "fake_orig", "fake_composite",
               "patch", "fake"}
_PATTERN = re.compile(r"(\w*)\((\w*)(,[a-zA-Z0-9_,=\.]*)?\)")


def _split(s: str, delim: str) -> List[str]:
    return [x for x in s.strip().strip("'").replace(" ", "").split(delim) if len(x) > 0]


def compute_iou(source: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """losses.py:649-666 (soft IoU loss; per-sample when the inputs are B x H x W)."""
    assert source.shape == target.shape
    eps = 1e-8
    if source.ndim == 3:
        inter = torch.sum(source * target, dim=(1, 2))
        union = torch.sum(source + target, dim=(1, 2)) - inter + eps
        return 1.0 - (inter / union).mean()
    inter = torch.sum(source * target)
    union = torch.sum(source + target) - inter + eps
    return 1.0 - (inter / union)


def compute_dice(source: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """losses.py:634-646."""
    assert source.ndim == 3 and source.shape == target.shape
    eps = 1e-8
    inter = torch.sum(source * target, dim=(1, 2))
    total = torch.sum(source.pow(2) + target.pow(2), dim=(1, 2)) + eps
    return 1.0 - 2.0 * (inter / total).mean()


class LossItem:
    """One ``<name>(<component>)`` term (losses.py:236-338)."""

    def __init__(self, name: str, component: str, **args):
        self.name, self.component, self.args = name, component, args
        self.partial_loss_with_triband_input = False
        if name == "rgb":
            self.rgb = torch.tensor([float(args.get("r", 0.5)), float(args.get("g", 0.5)), float(args.get("b", 0.5))])
            self.loss_name = args.get("loss", "L1")
            self.mean_rgb = bool(args.get("mean_rgb", False))
            if self.loss_name not in ("L1", "L2"):
                raise RuntimeError(f"Unknown loss name {self.loss_name}")

    def full_name(self) -> str:
        return f"{self.name}_{self.component}"

    def _unsupported(self):
        raise RuntimeError(f"Unsupported component for {self.component} for loss {self.name}")

    def _prep(self, value, truth):                                   # losses.py:24-34, 249-253
        if self.partial_loss_with_triband_input:
            mask = torch.logical_or(truth < 0.1, truth > 0.9)
            return value[mask], truth[mask]
        return value, truth

    def _foreground(self, d):                                        # losses.py:295-312 (U primary, V secondary, S canvas)
        if self.component == "uvs":
            return torch.sum(d["uvs"][:, :2, ...], dim=1)
        if self.component == "u":
            return d["uvs"][:, 0, ...]
        if self.component == "alpha":
            return d["alpha"][:, 0, ...]
        self._unsupported()

    def _background(self, d):                                        # losses.py:314-324
        if self.component == "uvs":
            return d["uvs"][:, 2, ...]
        if self.component == "alpha":
            return d["alpha"][:, 1, ...]
        self._unsupported()

    def compute(self, d: Dict[str, torch.Tensor], geom_truth) -> torch.Tensor:
        n = self.name
        if n in ("iou", "dice"):                                     # losses.py:453-462, 477-486
            src, tgt = self._prep(self._foreground(d), 1 - geom_truth.squeeze(1))
            return compute_iou(src, tgt) if n == "iou" else compute_dice(src, tgt)
        if n in ("iou_inv", "dice_inv"):                             # losses.py:465-474, 489-498
            src, tgt = self._prep(self._background(d), geom_truth.squeeze(1))
            return compute_iou(src, tgt) if n == "iou_inv" else compute_dice(src, tgt)
        if n == "l1":                                                # losses.py:501-530
            c = self.component
            if c == "fake_img":
                tgt, src = d["fake_img"].detach(), d["fake_img"]
            elif c == "fake_orig":
                tgt, src = d["fake_orig"].detach(), d["fake_img"]
            elif c == "fake_composite":
                tgt, src = d["fake"], d["fake_composite"]
            elif c == "patch":
                tgt, src = d["patch1"], d["patch2"]
            elif c == "canvas":
                raise RuntimeError("l1(canvas) belongs to the 'canvas' colour format (random crops); this build ships 'triad'")
            else:
                src, tgt = self._prep(self._foreground(d), 1 - geom_truth.squeeze(1))
            return torch.nn.functional.l1_loss(src, tgt)
        if n == "gan":                                               # losses.py:533-546
            key = f"{self.component}_logits"
            if key not in d:
                raise RuntimeError(f"Key {key} expected in: {list(d.keys())}")
            return torch.nn.functional.softplus(-d[key]).mean()
        if n == "rgb":                                               # losses.py:341-377
            c = self.component
            if c == "uvs":
                x = d["uvs"] * 2 - 1
            elif c in ("color_0", "color_1", "color_2"):
                x = d["colors"][..., int(c[-1])]
            else:
                self._unsupported()
            x = x * 0.5 + 0.5
            if self.mean_rgb:
                x = torch.stack([x[:, 0].mean(), x[:, 1].mean(), x[:, 2].mean()])
            shp = [1] * x.ndim
            shp[1 if len(shp) > 1 else 0] = 3
            tgt = self.rgb.to(x.device, x.dtype).reshape(*shp).expand_as(x)
            return torch.nn.functional.l1_loss(x, tgt) if self.loss_name == "L1" else torch.nn.functional.mse_loss(x, tgt)
        raise RuntimeError(f"Loss {n} not found in registered losses: " + ", ".join(sorted(_ITEMS)))


_ITEMS = {"iou", "iou_inv", "dice", "dice_inv", "l1", "gan", "rgb"}
_NEEDS_LPIPS = {"lpips", "plpips"}


def parse_loss_item(config: str) -> Tuple[float, LossItem]:
    """``<float>*<loss_name>(<component>[,arg=val...])`` (losses.py:146-233)."""
    parts = _split(config, "*")
    if len(parts) == 1:
        weight = 1.0
    elif len(parts) == 2:
        weight = float(parts[0])
    else:
        raise RuntimeError(f"Mis-configured loss string {config}")
    m = re.match(_PATTERN, parts[-1])
    if m is None:
        raise RuntimeError(f"Mis-configured loss string {config}; expected pattern <float>*<loss_name>(<component>)")
    name, component, argstr = m.group(1), m.group(2), m.group(3)
    if name in _NEEDS_LPIPS:
        raise RuntimeError(f"Loss {name} needs the LPIPS network, which is not part of this build (nor of the reference tree)")
    if name not in _ITEMS:
        raise RuntimeError(f"Loss {name} not found in registered losses: " + ", ".join(sorted(_ITEMS)))
    if component not in _COMPONENTS:
        raise RuntimeError(f'Component "{component}" not in valid values: ' + ", ".join(sorted(_COMPONENTS)))
    args = {}
    for part in _split(argstr or "", ","):
        kv = _split(part, "=")
        assert len(kv) == 2 and kv[0] not in args, f"Invalid argument string {argstr}"
        args[kv[0]] = kv[1]
    return weight, LossItem(name, component, **args)


class ForgerLosses:
    """Weighted sum of loss items (losses.py:37-119)."""

    def __init__(self, losses: List[LossItem], weights: List[float]):
        self.losses, self.weights = losses, weights
        names = [l.full_name() for l in losses]
        assert len(losses) == len(weights)
        for nm in names:
            if names.count(nm) > 1:
                raise RuntimeError(f"Loss with identifier {nm} defined more than once")

    @staticmethod
    def create_from_string(config: str) -> "ForgerLosses":
        items = [parse_loss_item(x) for x in _split(config or "", "+")]
        return ForgerLosses([i[1] for i in items], [i[0] for i in items])

    def set_partial_loss_with_triband_input(self, val: bool):
        for l in self.losses:
            l.partial_loss_with_triband_input = val

    def require_original_fake_image(self) -> bool:
        return any(l.component == "fake_orig" for l in self.losses)

    def is_empty(self) -> bool:
        return len(self.losses) == 0

    def compute(self, raw: Dict[str, torch.Tensor], geom_truth):
        total, results = 0, {}
        for l, w in zip(self.losses, self.weights):
            results[l.full_name()] = l.compute(raw, geom_truth)
            total = total + w * results[l.full_name()]
        return total, results


# ------------------------------------------------------------------------------------------------
# stitching (forger/train/stitching.py:28-267)
# ------------------------------------------------------------------------------------------------
class _Area:
    def __init__(self, rstart, cstart, rend, cend):
        self.rstart, self.cstart, self.rend, self.cend = rstart, cstart, rend, cend
        self.min_width = min(rend - rstart, cend - cstart)           # negative if there is no overlap


class CropHelper:
    """Crops are (row_start, col_start, height, width) on the full drawing."""

    @staticmethod
    def position_delta(crop1, crop2) -> torch.Tensor:
        return torch.tensor([crop2[0] - crop1[0], crop2[1] - crop1[1]], dtype=torch.int64)

    @staticmethod
    def compute_absolute_overlap(a, b) -> _Area:
        return _Area(max(a[0], b[0]), max(a[1], b[1]), min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3]))

    @staticmethod
    def compute_overlaps(a, b):
        ov = CropHelper.compute_absolute_overlap(a, b)
        if ov.min_width <= 0:
            return ov, None, None
        rel = lambda c: _Area(ov.rstart - c[0], ov.cstart - c[1], ov.rend - c[0], ov.cend - c[1])
        return ov, rel(a), rel(b)

    @staticmethod
    def offset_crop(crop, margin):
        return (crop[0] + margin, crop[1] + margin, crop[2] - 2 * margin, crop[3] - 2 * margin)

    @staticmethod
    def composite(im1, im2, area1: _Area, area2: _Area, alpha1=None):
        """im1 with the pixels of area2 of im2 in area1 (optionally blended with alpha1), stitching.py:161-179."""
        mask1 = torch.ones_like(im1[:1, :1, ...])
        mask1[..., area1.rstart:area1.rend, area1.cstart:area1.cend] = alpha1 if alpha1 is not None else 0
        res = mask1 * im1
        res[..., area1.rstart:area1.rend, area1.cstart:area1.cend] += \
            (1 - alpha1 if alpha1 is not None else 1.0) * im2[..., area2.rstart:area2.rend, area2.cstart:area2.cend]
        return res

    @staticmethod
    def gen_overlapping_square_crop(input_width, crop1, margin, min_overlap, rng=random):
        width = crop1[2]
        radius = width - margin - min_overlap - 1
        ij = [0, 0]
        for x in range(2):
            rmin = max(0, crop1[x] - radius)
            rmax = min(crop1[x] + radius, input_width - width - 1)
            ij[x] = rng.randint(rmin, rmax)
        return ij[0], ij[1], width, width


class RandomStitcher:
    """stitching.py:194-267."""

    def __init__(self, crop_margin: int = 10, min_overlap: int = 50):
        self.margin, self.min_overlap = crop_margin, min_overlap

    def gen_overlapping_square_crop(self, input_width, crop1, rng=random):
        return CropHelper.gen_overlapping_square_crop(input_width, crop1, self.margin, self.min_overlap, rng)

    def offset_crop(self, crop):
        return CropHelper.offset_crop(crop, self.margin)

    @staticmethod
    def gen_random_positions(batch, width):
        return torch.randint(0, width - 1, (batch, 2))

    def generate_with_stitching(self, G, z, c, geom_feature1, geom_feature2, crop1, crop2, positions1=None, **g_kwargs):
        """Two generator passes over two overlapping crops of one drawing, with noise positions that differ by the crop
        offset, and each result composited into the other where the (margin-inset) crops overlap."""
        res = G.img_resolution
        if positions1 is None:
            positions1 = self.gen_random_positions(z.shape[0], width=res).to(z.device)
        positions2 = positions1 + CropHelper.position_delta(crop1, crop2).unsqueeze(0).to(z.device)
        fake1 = G(z, c, geom_feature1, positions=positions1, style_mixing_prob=0, **g_kwargs)
        fake2 = G(z, c, geom_feature2, positions=positions2, style_mixing_prob=0, **g_kwargs)
        _, area1, area2 = CropHelper.compute_overlaps(crop1, self.offset_crop(crop2))
        fake1_composite = CropHelper.composite(fake1, fake2, area1, area2)
        _, area1, area2 = CropHelper.compute_overlaps(self.offset_crop(crop1), crop2)
        fake2_composite = CropHelper.composite(fake2, fake1, area2, area1)
        return {"fake1": fake1, "fake2": fake2, "fake1_composite": fake1_composite, "fake2_composite": fake2_composite,
                "positions1": positions1, "positions2": positions2,
                "patch1": fake1[..., area1.rstart:area1.rend, area1.cstart:area1.cend],
                "patch2": fake2[..., area2.rstart:area2.rend, area2.cstart:area2.cend]}
