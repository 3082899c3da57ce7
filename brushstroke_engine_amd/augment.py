"""Adaptive discriminator augmentation pipeline (reference: ``training/augment.py:112-431``, the
"Training Generative Adversarial Networks with Limited Data" pipeline) on this build's operators.

Same constructor arguments, the same ``p`` buffer (overall probability multiplier that the training loop adapts), the same
``forward(images, debug_percentile=None)`` and the same order of stages:

  pixel blitting (x-flip, 90-degree rotations, integer translation) and general geometric transforms (isotropic scale,
  pre-rotation, anisotropic scale, post-rotation, fractional translation) are composed into ONE inverse homogeneous 2-D
  transform per sample and executed once: reflect padding, 2x up-sampling with the sym6 low-pass filter
  (``ops.upsample2d`` = the HIP ``upfirdn2d``), ``affine_grid`` + bilinear ``grid_sample``, 2x down-sampling with the
  same filter; colour transforms (brightness, contrast, luma flip, hue rotation, saturation) are composed into one
  homogeneous 3-D colour transform per sample; then per-sample band amplification (separable 1-D filters through
  ``ops.upfirdn2d``), additive noise and cutout.

``debug_percentile`` replaces every random draw by its value at that percentile, which makes the pipeline deterministic
(the golden vectors use it).  ``grid_sample`` is wrapped so that it can be differentiated twice w.r.t. the images (the R1
penalty back-propagates through the augmented real images; torch has no derivative for ``grid_sampler_2d_backward``).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops

# Published orthogonal wavelet low-pass decomposition filters (Symlets 2 and 6)
_SYM2 = [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025]
_SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
         0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
         0.0017677118642428036, -0.007800708325034148]


class _GridSample(torch.autograd.Function):
    """Bilinear ``grid_sample`` (zero padding, align_corners=False), twice differentiable w.r.t. the input."""

    @staticmethod
    def forward(ctx, x, grid):
        ctx.save_for_backward(x, grid)
        return torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)

    @staticmethod
    def backward(ctx, dy):
        x, grid = ctx.saved_tensors
        return _GridSampleInputGrad.apply(dy, x, grid), None


class _GridSampleInputGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, grid):
        ctx.save_for_backward(grid)
        gx, _ = torch.ops.aten.grid_sampler_2d_backward(dy, x, grid, 0, 0, False, (True, False))
        return gx

    @staticmethod
    def backward(ctx, ggx):
        grid, = ctx.saved_tensors
        # the input gradient is linear in dy (and does not depend on x): its adjoint is the sampling itself
        return _GridSample.apply(ggx, grid), None, None


def _stack_matrix(rows, like: Optional[torch.Tensor], device):
    """rows of python floats / [B] tensors -> [B, r, c] (or [r, c] when every entry is a constant)."""
    flat = [e for row in rows for e in row]
    tens = [e for e in flat if torch.is_tensor(e)]
    if not tens:
        return torch.tensor(rows, dtype=torch.float32, device=device)
    shape = tens[0].shape
    # constants first (one small tensor built from python floats), then the per-sample entries written into their slots
    r, c = len(rows), len(rows[0])
    base = torch.tensor([0.0 if torch.is_tensor(e) else float(e) for e in flat], dtype=torch.float32, device=device)
    out = base.expand(tuple(shape) + (r * c,)).clone()
    for k, e in enumerate(flat):
        if torch.is_tensor(e):
            out[..., k] = e
    return out.reshape(tuple(shape) + (r, c))


def _upload(t: torch.Tensor, device) -> torch.Tensor:
    """Host tensor -> device through page-locked memory: an asynchronous copy in stream order (a pageable source makes the copy
    wait for the stream to drain, i.e. a host/device sync per augmentation call)."""
    return t.contiguous().pin_memory().to(device, non_blocking=True)


class _OneHostThread:
    """The host-side parameter math below works on tensors of a few dozen floats; torch's CPU operators would fan each of them
    out over the intra-op thread pool (measured: 1.3 ms for a 64-element max with 8 threads, 17 us with one)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *exc):
        torch.set_num_threads(self.n)


class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125,
                 scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125,
                 brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5, hue_max=1,
                 saturation_std=1, imgfilter=0, imgfilter_bands: Sequence[float] = (1, 1, 1, 1), imgfilter_std=1,
                 noise=0, cutout=0, noise_std=0.1, cutout_size=0.5):
        super().__init__()
        self.register_buffer("p", torch.ones([]))
        for k, v in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate, aniso=aniso,
                         xfrac=xfrac, scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std, xfrac_std=xfrac_std,
                         brightness=brightness, contrast=contrast, lumaflip=lumaflip, hue=hue, saturation=saturation,
                         brightness_std=brightness_std, contrast_std=contrast_std, hue_max=hue_max, saturation_std=saturation_std,
                         imgfilter=imgfilter, imgfilter_std=imgfilter_std, noise=noise, cutout=cutout, noise_std=noise_std,
                         cutout_size=cutout_size).items():
            setattr(self, k, float(v))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer("Hz_geom", ops.setup_filter(_SYM6))                 # 12 taps -> separable 1-D filter
        # band-pass filter bank of the image-space filtering stage: octave bands of the sym2 half-band pair
        lo = np.asarray(_SYM2)
        hi = lo * ((-1) ** np.arange(lo.size))
        lo2 = np.convolve(lo, lo[::-1]) / 2
        hi2 = np.convolve(hi, hi[::-1]) / 2
        bank = np.eye(4, 1)
        for i in range(1, 4):
            up = np.zeros((4, bank.shape[1] * 2 - 1))
            up[:, ::2] = bank                                                    # zero-stuff: next octave
            bank = np.stack([np.convolve(row, lo2) for row in up])
            c = bank.shape[1] // 2
            bank[i, c - hi2.size // 2: c - hi2.size // 2 + hi2.size] += hi2
        self.register_buffer("Hz_fbank", torch.as_tensor(bank, dtype=torch.float32))
        self._fbank_host = torch.as_tensor(bank, dtype=torch.float32)          # host copy for the per-sample tap mixing

    # -- random choices ----------------------------------------------------------------------------------------------
    def _p_value(self) -> float:
        """The augmentation probability as a python float: read from the device buffer only when it has changed (in-place
        writes bump the tensor's version counter)."""
        c = getattr(self, "_p_cache", None)
        if c is None or c[0] is not self.p or c[1] != self.p._version:
            c = self._p_cache = (self.p, self.p._version, float(self.p))
        return c[2]

    def _gate(self, value, prob, neutral, shape, dev):
        """``value`` where a uniform draw falls below prob * p, ``neutral`` elsewhere."""
        keep = torch.rand(shape, device=dev) < prob * self._p_value()
        return torch.where(keep, value, torch.full_like(value, neutral))

    def forward(self, images: torch.Tensor, debug_percentile=None) -> torch.Tensor:
        with _OneHostThread():
            return self._forward(images, debug_percentile)

    def _forward(self, images: torch.Tensor, debug_percentile=None) -> torch.Tensor:
        assert torch.is_tensor(images) and images.ndim == 4
        B, C, H, W = images.shape
        img_dev = images.device
        # The per-sample random parameters and the 3x3 / 4x4 matrices composed from them are a few hundred scalars: they are
        # drawn and multiplied on the HOST (torch CPU ops, no launches) and only the finished matrices travel to the device --
        # the reference builds them from dozens of tiny device kernels and then reads the padding margins back
        # (augment.py:262-263, a device-to-host sync per call); here nothing waits for the device.
        dev = torch.device("cpu")
        p_now = self._p_value()
        dp = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32).cpu()
        normal_q = (lambda: torch.erfinv(dp * 2 - 1)) if dp is not None else None         # N(0,1)/sqrt(2) quantile as the reference uses it
        M = lambda rows: _stack_matrix(rows, None, dev)

        # ---- geometry: inverse transform G (output pixel -> input pixel), composed right to left ----
        G = None

        def compose(m):
            nonlocal G
            G = m if G is None else G @ m
        if self.xflip > 0:
            i = self._gate(torch.floor(torch.rand([B], device=dev) * 2), self.xflip, 0.0, [B], dev)
            if dp is not None:
                i = torch.full_like(i, float(torch.floor(dp * 2)))
            compose(M([[1 / (1 - 2 * i), 0, 0], [0, 1, 0], [0, 0, 1]]))
        if self.rotate90 > 0:
            i = self._gate(torch.floor(torch.rand([B], device=dev) * 4), self.rotate90, 0.0, [B], dev)
            if dp is not None:
                i = torch.full_like(i, float(torch.floor(dp * 4)))
            th = math.pi / 2 * i                                                  # inverse of a rotation by -pi/2 * i
            compose(M([[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]]))
        if self.xint > 0:
            t = (torch.rand([B, 2], device=dev) * 2 - 1) * self.xint_max
            t = torch.where(torch.rand([B, 1], device=dev) < self.xint * p_now, t, torch.zeros_like(t))
            if dp is not None:
                t = torch.full_like(t, float((dp * 2 - 1) * self.xint_max))
            compose(M([[1, 0, -torch.round(t[:, 0] * W)], [0, 1, -torch.round(t[:, 1] * H)], [0, 0, 1]]))
        if self.scale > 0:
            s = self._gate(torch.exp2(torch.randn([B], device=dev) * self.scale_std), self.scale, 1.0, [B], dev)
            if dp is not None:
                s = torch.full_like(s, float(torch.exp2(normal_q() * self.scale_std)))
            compose(M([[1 / s, 0, 0], [0, 1 / s, 0], [0, 0, 1]]))
        p_rot = 1 - math.sqrt(min(max(1 - self.rotate * p_now, 0.0), 1.0))          # P(pre OR post) = rotate * p

        def rotation(debug_value):
            th = (torch.rand([B], device=dev) * 2 - 1) * math.pi * self.rotate_max
            th = torch.where(torch.rand([B], device=dev) < p_rot, th, torch.zeros_like(th))
            if dp is not None:
                th = torch.full_like(th, debug_value)
            compose(M([[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]]))
        if self.rotate > 0:
            rotation(float((dp * 2 - 1) * math.pi * self.rotate_max) if dp is not None else 0.0)
        if self.aniso > 0:
            s = self._gate(torch.exp2(torch.randn([B], device=dev) * self.aniso_std), self.aniso, 1.0, [B], dev)
            if dp is not None:
                s = torch.full_like(s, float(torch.exp2(normal_q() * self.aniso_std)))
            compose(M([[1 / s, 0, 0], [0, s, 0], [0, 0, 1]]))
        if self.rotate > 0:
            rotation(0.0)
        if self.xfrac > 0:
            t = torch.randn([B, 2], device=dev) * self.xfrac_std
            t = torch.where(torch.rand([B, 1], device=dev) < self.xfrac * p_now, t, torch.zeros_like(t))
            if dp is not None:
                t = torch.full_like(t, float(normal_q() * self.xfrac_std))
            compose(M([[1, 0, -t[:, 0] * W], [0, 1, -t[:, 1] * H], [0, 0, 1]]))
        if G is not None:
            images = self._warp(images, G if G.ndim == 3 else G.expand(B, 3, 3), dev)      # (G on the host)

        # ---- colour: homogeneous 4x4 transform, composed left to right (later stages multiply from the left) ----
        Cm = None

        def ccompose(m):
            nonlocal Cm
            Cm = m if Cm is None else m @ Cm
        eye4 = torch.eye(4, device=dev)
        v = torch.tensor([1, 1, 1, 0], dtype=torch.float32, device=dev) / math.sqrt(3)      # luma axis
        vv = torch.outer(v, v)
        if self.brightness > 0:
            b = self._gate(torch.randn([B], device=dev) * self.brightness_std, self.brightness, 0.0, [B], dev)
            if dp is not None:
                b = torch.full_like(b, float(normal_q() * self.brightness_std))
            ccompose(M([[1, 0, 0, b], [0, 1, 0, b], [0, 0, 1, b], [0, 0, 0, 1]]))
        if self.contrast > 0:
            c = self._gate(torch.exp2(torch.randn([B], device=dev) * self.contrast_std), self.contrast, 1.0, [B], dev)
            if dp is not None:
                c = torch.full_like(c, float(torch.exp2(normal_q() * self.contrast_std)))
            ccompose(M([[c, 0, 0, 0], [0, c, 0, 0], [0, 0, c, 0], [0, 0, 0, 1]]))
        if self.lumaflip > 0:
            i = torch.floor(torch.rand([B, 1, 1], device=dev) * 2)
            i = torch.where(torch.rand([B, 1, 1], device=dev) < self.lumaflip * p_now, i, torch.zeros_like(i))
            if dp is not None:
                i = torch.full_like(i, float(torch.floor(dp * 2)))
            ccompose(eye4 - 2 * vv * i)                                            # Householder reflection about the luma axis
        if self.hue > 0 and C > 1:
            th = self._gate((torch.rand([B], device=dev) * 2 - 1) * math.pi * self.hue_max, self.hue, 0.0, [B], dev)
            if dp is not None:
                th = torch.full_like(th, float((dp * 2 - 1) * math.pi * self.hue_max))
            s_, c_ = torch.sin(th), torch.cos(th)
            k = 1 - c_
            x, y, z = float(v[0]), float(v[1]), float(v[2])
            ccompose(M([[x * x * k + c_, x * y * k - z * s_, x * z * k + y * s_, 0],
                        [y * x * k + z * s_, y * y * k + c_, y * z * k - x * s_, 0],
                        [z * x * k - y * s_, z * y * k + x * s_, z * z * k + c_, 0],
                        [0, 0, 0, 1]]))
        if self.saturation > 0 and C > 1:
            s = torch.exp2(torch.randn([B, 1, 1], device=dev) * self.saturation_std)
            s = torch.where(torch.rand([B, 1, 1], device=dev) < self.saturation * p_now, s, torch.ones_like(s))
            if dp is not None:
                s = torch.full_like(s, float(torch.exp2(normal_q() * self.saturation_std)))
            ccompose(vv + (eye4 - vv) * s)
        if Cm is not None:
            if Cm.ndim == 2:
                Cm = Cm.expand(B, 4, 4)
            Cm = _upload(Cm, img_dev)
            flat = images.reshape(B, C, H * W)
            if C == 3:
                flat = Cm[:, :3, :3] @ flat + Cm[:, :3, 3:]
            elif C == 1:
                row = Cm[:, :3, :].mean(dim=1, keepdim=True)
                flat = flat * row[:, :, :3].sum(dim=2, keepdim=True) + row[:, :, 3:]
            else:
                raise ValueError("Image must be RGB (3 channels) or L (1 channel)")
            images = flat.reshape(B, C, H, W)

        # ---- image-space filtering: per-sample amplification of four octave bands ----
        if self.imgfilter > 0:
            nb = self.Hz_fbank.shape[0]
            assert len(self.imgfilter_bands) == nb
            power = torch.tensor([10, 1, 1, 1], dtype=torch.float32, device=dev) / 13       # expected 1/f power per band
            gain = torch.ones([B, nb], device=dev)
            for i, strength in enumerate(self.imgfilter_bands):
                t_i = self._gate(torch.exp2(torch.randn([B], device=dev) * self.imgfilter_std), self.imgfilter * strength, 1.0, [B], dev)
                if dp is not None:
                    t_i = torch.full_like(t_i, float(torch.exp2(normal_q() * self.imgfilter_std))) if strength > 0 else torch.ones_like(t_i)
                t = torch.ones([B, nb], device=dev)
                t[:, i] = t_i
                gain = gain * (t / (power * t.square()).sum(dim=-1, keepdim=True).sqrt())
            taps = _upload(gain @ self._fbank_host, img_dev)                        # [B, taps]: one separable filter per sample
            pad = self.Hz_fbank.shape[1] // 2
            x = torch.nn.functional.pad(images, [pad, pad, pad, pad], mode="reflect")
            outs = []
            for n in range(B):                                                     # (a different filter per sample)
                outs.append(ops.upfirdn2d(x[n:n + 1].contiguous(), taps[n].contiguous(), flip_filter=True))
            images = torch.cat(outs, dim=0)

        # ---- corruptions ----
        if self.noise > 0:
            sigma = torch.randn([B, 1, 1, 1], device=dev).abs() * self.noise_std
            sigma = torch.where(torch.rand([B, 1, 1, 1], device=dev) < self.noise * p_now, sigma, torch.zeros_like(sigma))
            if dp is not None:
                sigma = torch.full_like(sigma, float(torch.erfinv(dp) * self.noise_std))
            images = images + torch.randn([B, C, H, W], device=img_dev) * _upload(sigma, img_dev)
        if self.cutout > 0:
            size = torch.full([B, 2, 1, 1, 1], self.cutout_size, device=dev)
            size = torch.where(torch.rand([B, 1, 1, 1, 1], device=dev) < self.cutout * p_now, size, torch.zeros_like(size))
            center = torch.rand([B, 2, 1, 1, 1], device=dev)
            if dp is not None:
                size = torch.full_like(size, self.cutout_size)
                center = torch.full_like(center, float(dp))
            size, center = _upload(size, img_dev), _upload(center, img_dev)
            cx = (torch.arange(W, device=img_dev).reshape(1, 1, 1, -1) + 0.5) / W
            cy = (torch.arange(H, device=img_dev).reshape(1, 1, -1, 1) + 0.5) / H
            outside = torch.logical_or((cx - center[:, 0]).abs() >= size[:, 0] / 2, (cy - center[:, 1]).abs() >= size[:, 1] / 2)
            images = images * outside.to(torch.float32)
        return images

    # -- execution of the geometric transform ------------------------------------------------------------------------
    def _warp(self, images, G, dev):
        B, C, H, W = images.shape
        cx, cy = (W - 1) / 2, (H - 1) / 2
        # how far the transformed image corners reach decides the reflect padding
        corners = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32, device=dev)
        cp = G @ corners.t()                                                       # [B, 3, 4]
        hz_pad = self.Hz_geom.shape[0] // 4
        m = cp[:, :2, :].permute(1, 0, 2).flatten(1)                               # [xy, B*4]
        m = torch.cat([-m, m]).max(dim=1).values                                   # [x0, y0, x1, y1]
        m = m + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=torch.float32, device=dev)
        m = m.max(torch.zeros(4, device=dev)).min(torch.tensor([W - 1, H - 1] * 2, dtype=torch.float32, device=dev))
        mx0, my0, mx1, my1 = (int(t) for t in m.ceil().to(torch.int32))           # (host tensors: no device sync)
        images = torch.nn.functional.pad(images, [mx0, mx1, my0, my1], mode="reflect")
        T = lambda tx, ty: torch.tensor([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=torch.float32, device=dev)
        S = lambda sx, sy: torch.tensor([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], dtype=torch.float32, device=dev)
        G = T((mx0 - mx1) / 2, (my0 - my1) / 2) @ G
        images = ops.upsample2d(images.contiguous(), self.Hz_geom, up=2)
        G = S(2, 2) @ G @ S(0.5, 0.5)
        G = T(-0.5, -0.5) @ G @ T(0.5, 0.5)
        shape = [B, C, (H + hz_pad * 2) * 2, (W + hz_pad * 2) * 2]
        G = S(2 / images.shape[3], 2 / images.shape[2]) @ G @ S(shape[3] / 2, shape[2] / 2)
        grid = torch.nn.functional.affine_grid(theta=_upload(G[:, :2, :], images.device), size=shape, align_corners=False)
        images = _GridSample.apply(images, grid)
        return ops.downsample2d(images.contiguous(), self.Hz_geom, down=2, padding=-hz_pad * 2, flip_filter=True)
