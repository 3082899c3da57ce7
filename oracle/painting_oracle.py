"""CPU restatement of the painting-engine driver around the generator -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module; the product
path (``brushstroke_engine_amd/painting.py``) never does.  It walks the canvas tile by tile in the reference's order
-- one ``render_stroke`` per tile, feature canvas updated in between -- which is exactly what the product's batched
three-phase schedule must reproduce.

Pinned by ``tests/golden/engine_r128.npz`` (made by ``tests/golden/make_golden_engine.py`` from the reference's own
``PaintEngineFactory`` / ``PaintingHelper`` run over a 9-tile canvas, blending levels 0 and 2): tile list, padded
geometry, encoder features, the feature canvas and both RGBA canvases.

Reference:
  geometry encoder      forger/experimental/autoenc/simple_autoencoder.py:88-121, 155-199, 251-261; base.py:123-134
  prepare_geom_input    forger/ui/brush.py:672-681
  dirty-area alpha      forger/ui/brush.py:159-187
  blended features      forger/ui/brush.py:190-227, 239-242; FeatureCanvas :33-92
  render_stroke         forger/ui/brush.py:244-398
  triad compositing     forger/ui/brush.py:763-792
  tiling / paste        forger/viz/style_transfer.py:15-48; forger/viz/paint_image_main.py:58-61, 145-192
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import neube_oracle as no


# ------------------------------------------------------------------------------------------------
# geometry encoder (functional, straight from the state dict)
# ------------------------------------------------------------------------------------------------
def _conv_bn_lrelu(x, sd, prefix, stride, pad):
    """SingleConvolution: reflect-padded conv -> eval-mode BatchNorm (eps 1e-5) -> LeakyReLU(0.01)."""
    x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    x = F.conv2d(x, sd[prefix + ".conv.0.weight"], sd[prefix + ".conv.0.bias"], stride=stride)
    x = F.batch_norm(x, sd[prefix + ".conv.1.running_mean"], sd[prefix + ".conv.1.running_var"],
                     sd[prefix + ".conv.1.weight"], sd[prefix + ".conv.1.bias"], training=False, eps=1e-5)
    return F.leaky_relu(x, 0.01)


def encoder_encode(esd: Dict[str, np.ndarray], geom: torch.Tensor, preproc_type=None, resolutions=(0, 1)) -> List[torch.Tensor]:
    """``AutoEncoder.encode(geom, res=[0,1])`` for the default ``sauto`` flags."""
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in esd.items()}
    x = geom.to(torch.float32)
    if preproc_type == "-11inverse":
        x = (1 - x) * 2 - 1
    elif preproc_type == "inverse":
        x = 1 - x
    elif preproc_type not in (None, "none"):
        raise RuntimeError(f'Unknown preprocessing type "{preproc_type}"')
    x = _conv_bn_lrelu(x, sd, "encoder.model.0", 1, 3)           # 7x7 stem
    for i in (1, 2, 3):
        x = _conv_bn_lrelu(x, sd, f"encoder.model.{i}", 2, 1)    # three stride-2 stages
    for i in (4, 5):
        x = _conv_bn_lrelu(x, sd, f"encoder.model.{i}", 1, 1)    # bottleneck 256 -> 32 -> 16
    results = [x]
    for i in range(max(resolutions)):                            # decode_partial
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        x = _conv_bn_lrelu(x, sd, f"decoder.model.{i}.conv", 1, 1)
        results.append(x)
    return [results[r] for r in resolutions]


# ------------------------------------------------------------------------------------------------
# tiling
# ------------------------------------------------------------------------------------------------
def pad_geo(geo: np.ndarray, crop_margin: int) -> np.ndarray:
    out = np.full((geo.shape[0] + crop_margin, geo.shape[1] + crop_margin, geo.shape[2]), 255, np.uint8)
    out[crop_margin:, crop_margin:, :] = geo
    return out


def generate_stitching_crops(stroke_image: np.ndarray, patch_width: int, mode="all", overlap_margin=15):
    rwidth = patch_width - overlap_margin * 2
    h, w, ch = stroke_image.shape
    nrows, ncols = h // rwidth + 1, w // rwidth + 1
    padded = np.full((nrows * rwidth + patch_width, ncols * rwidth + patch_width, ch), 255, np.uint8)
    padded[:h, :w] = stroke_image
    crops = []
    for r in range(nrows):
        for c in range(ncols):
            y, x = r * rwidth, c * rwidth
            if mode == "all" or np.sum(padded[y:y + patch_width, x:x + patch_width] < 0.001) > 10:
                crops.append((y, x, patch_width, patch_width))
    return crops, padded


def dirty_area_alpha(width: int, margin: int, crop_margin: int = 0) -> torch.Tensor:
    """``generate_dirty_area_alpha`` for a dirty area that spans the whole tile (brush.py:160-164 insets it)."""
    r0 = c0 = margin + crop_margin
    r1 = c1 = r0 + width - 2 * margin - 2 * crop_margin
    x = torch.linspace(0, width - 1, steps=width)
    gy, gx = torch.meshgrid(x, x, indexing="ij")
    dx = torch.min(torch.pow(gx - c0, 2), torch.pow(gx - c1 + 1, 2))
    dy = torch.min(torch.pow(gy - r0, 2), torch.pow(gy - r1 + 1, 2))
    d = dx + dy
    d[0:r0, c0:c1] = dy[0:r0, c0:c1]
    d[r1:, c0:c1] = dy[r1:, c0:c1]
    d[r0:r1, 0:c0] = dx[r0:r1, 0:c0]
    d[r0:r1, c1:] = dx[r0:r1, c1:]
    res = 1 - torch.sqrt(d) / margin
    res[res < 0] = 0
    res[r0:r1, c0:c1] = 1
    return res


class OraclePainter:
    """Sequential ``PaintingHelper`` + ``TriadGanPaintEngine`` restatement over an ``OracleGenerator``."""

    feature_blending_margin = 16

    def __init__(self, gen: "no.OracleGenerator", encoder_sd: Dict[str, np.ndarray], preproc_type=None):
        self.G, self.esd, self.preproc = gen, encoder_sd, preproc_type
        self.R = gen.cfg.img_resolution
        self.level = 0
        self.features = self.mask = None
        self.render_mode = "clear"
        self.sfactor = None            # set to enable UVS mapping (brush.py:773-774)

    def compute_sfactor(self, cal_medium: np.ndarray, cal_thick: np.ndarray, z=None, ws=None):
        """``StyleUVSMapper.get_sfactor`` (forger/ui/mapper.py:117-135) on caller-provided calibration drawings
        [5,R,R] uint8 (medium / thick strokes, 0 = stroke): 1 / (min over drawings of the 15th largest background
        weight S among the pixels that are background even in the thick version)."""
        geo = (torch.from_numpy(cal_medium).to(torch.float32) / 255).unsqueeze(1)
        bmask = (torch.from_numpy(cal_thick).to(torch.float32) / 255).unsqueeze(1) > 0.99
        feats = encoder_encode(self.esd, geo, self.preproc)
        n = geo.shape[0]
        if ws is not None:
            _, dbg = self.G.forward_pre_mapped(torch.as_tensor(ws).expand(n, -1, -1), feats, return_debug_data=True)
        else:
            _, dbg = self.G.forward(np.repeat(np.asarray(z), n, axis=0), None, feats, return_debug_data=True)
        S = dbg["uvs"][:, 2:3]
        val = torch.stack([torch.topk(S[i][bmask[i]], k=15)[0].min() for i in range(n)]).min()
        return 1 / val

    def make_new_canvas(self, rows, cols, feature_blending=0):
        self.level = feature_blending
        self.down = 2 ** (feature_blending - 1) if feature_blending > 0 else None
        self.rows, self.cols = rows, cols
        self.features = self.mask = None

    def render_stroke(self, stroke_patch: np.ndarray, z=None, ws=None, x=0, y=0, crop_margin=0, position=None,
                      user_colors=None):
        R = self.R
        geom = 1 - torch.from_numpy(stroke_patch[:, :, -1:]).to(torch.float32).permute(2, 0, 1) / 255.0
        geom = geom.unsqueeze(0)
        feats = encoder_encode(self.esd, geom, self.preproc)
        kw = {}
        upd = None
        if self.level > 0:
            df = self.down
            x, y = (x // df) * df, (y // df) * df
            bres = R // df
            margin, crop = self.feature_blending_margin // df, crop_margin // df
            ys, xs = y // df, x // df
            alpha = dirty_area_alpha(bres, margin, crop)
            upd = alpha > 0.99
            if self.mask is not None:
                m = self.mask[ys:ys + bres, xs:xs + bres]
                f = self.features[:, :, ys:ys + bres, xs:xs + bres]
                upd = upd | (m & (alpha > 0))
                alpha = alpha.clone()
                alpha[~m] = 1
                alpha = 1 - alpha
                kw["blended_features"] = {bres: {"features": f.clone(), "alpha": alpha[None, None]}}
            if crop > 0:
                upd[:crop, :] = False
                upd[-crop:, :] = False
                upd[:, :crop] = False
                upd[:, -crop:] = False
            kw["return_features"] = [bres]
        pos = None if position is None else np.asarray(position, np.int64).reshape(1, 2)
        if ws is not None:
            _, dbg = self.G.forward_pre_mapped(ws, feats, positions=pos, return_debug_data=True, **kw)
        else:
            _, dbg = self.G.forward(z, None, feats, positions=pos, return_debug_data=True, **kw)
        rgba = no.triad_composite(dbg["uvs"], dbg["colors"], self.render_mode, user_colors, self.sfactor)
        if self.level > 0:
            fnew = dbg[f"features{bres}"]
            if self.features is None:
                hc, wc = -(-self.rows // df), -(-self.cols // df)
                self.features = torch.zeros((1, fnew.shape[1], hc, wc), dtype=fnew.dtype)
                self.mask = torch.zeros((hc, wc), dtype=torch.bool)
            self.mask[ys:ys + bres, xs:xs + bres][upd] = True
            u4 = upd[None, None].expand(-1, fnew.shape[1], -1, -1)
            self.features[:, :, ys:ys + bres, xs:xs + bres][u4] = fnew[u4]
        img = no.rgba_to_uint8(rgba)[0].permute(1, 2, 0).numpy()
        if crop_margin > 0:
            img = img[crop_margin:R - crop_margin, crop_margin:R - crop_margin]
        return np.ascontiguousarray(img), {"x": x + crop_margin, "y": y + crop_margin}

    def paint_image(self, geom: np.ndarray, z=None, ws=None, crop_margin=10, feature_blending=0, render_mode="clear",
                    stitching_mode="all", user_colors=None, on_white=False):
        """``paint_image_main.py:145-192`` from the thresholded geometry image [H,W,1] (255 = background)."""
        R = self.R
        padded0 = pad_geo(geom, crop_margin)
        crops, padded = generate_stitching_crops(padded0, R, mode=stitching_mode, overlap_margin=2 * crop_margin)
        result = np.zeros((padded.shape[0], padded.shape[1], 4), np.uint8)
        self.make_new_canvas(result.shape[0], result.shape[1], feature_blending)
        self.render_mode = render_mode
        for (y, x, _, _) in crops:
            patch = 255 - padded[y:y + R, x:x + R, :]
            res, meta = self.render_stroke(patch, z=z, ws=ws, x=x, y=y, crop_margin=crop_margin, position=(y, x),
                                           user_colors=user_colors)
            result[meta["y"]:meta["y"] + res.shape[0], meta["x"]:meta["x"] + res.shape[1]] = res
        full = result
        if on_white:
            a = result[..., 3:].astype(np.float32) / 255
            result = (result[..., :3].astype(np.float32) * a + 255 * (1 - a)).clip(0, 255).astype(np.uint8)
        out = result[crop_margin:crop_margin + geom.shape[0], crop_margin:crop_margin + geom.shape[1], :]
        return out, full, crops, padded
