"""Operator layer: the reference's ``torch_utils/ops`` call surface on top of the HIP C ABI.

Same function names, argument meaning and error behaviour as the reference ops
(``bias_act.bias_act``, ``upfirdn2d.upfirdn2d`` / ``setup_filter``, ``networks.modulated_conv2d``),
forward only, fp32, device tensors only.  Every function launches hand-written gfx950 kernels
through :mod:`brushstroke_engine_amd._lib` on torch's current HIP stream; nothing here computes on
the CPU or falls back to torch ops.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

ACT_CODES = {"linear": 1, "relu": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5}   # = cuda_idx, bias_act.py:22-32
ACT_DEFAULTS = {"linear": (0.0, 1.0), "relu": (0.0, math.sqrt(2)), "lrelu": (0.2, math.sqrt(2)),
                "tanh": (0.0, 1.0), "sigmoid": (0.0, 1.0)}


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(x: torch.Tensor, name: str) -> None:
    if not isinstance(x, torch.Tensor):
        raise AssertionError(f"{name} must be a torch.Tensor")
    if x.device.type != "cuda":
        raise RuntimeError(f"{name} must reside on the GPU (got {x.device}); this build has no CPU path")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {x.dtype})")


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """Fused bias + activation + gain + clamp.  Mirrors ``bias_act.bias_act`` (bias_act.py:55-89)."""
    _dev(x, "x")
    if act not in ACT_CODES:
        raise AssertionError(f"unknown activation {act!r}")
    assert clamp is None or clamp >= 0
    d_alpha, d_gain = ACT_DEFAULTS[act]
    alpha = float(d_alpha if alpha is None else alpha)
    gain = float(d_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    x = x.contiguous()
    size_b, step_b = 0, 1
    if b is not None:
        _dev(b, "b")
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        b = b.contiguous()
        size_b, step_b = b.shape[0], x.stride(dim)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_bias_act_f32(_p(x), _p(b), _p(y), x.numel(), size_b, step_b, ACT_CODES[act], alpha,
                                              gain, clamp, _stream(x)), "bias_act")
    return y


def setup_filter(f: Sequence[float] = (1, 3, 3, 1), device=None, normalize=True, flip_filter=False, gain=1):
    """``upfirdn2d.setup_filter`` (upfirdn2d.py:72-116) for tap lists shorter than 8 (non-separable 2-D)."""
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in (1, 2) and f.numel() > 0
    if f.ndim == 1:
        assert f.numel() < 8, "separable filters are not used on the generator path"
        f = f.ger(f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip([0, 1])
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device) if device is not None else f


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return padding


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Pad, upsample, FIR-filter, downsample.  Mirrors ``upfirdn2d.upfirdn2d`` (upfirdn2d.py:120-164)."""
    _dev(x, "x")
    assert x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    _dev(f, "f")
    assert f.ndim == 2, "f must be rank 2"
    upx, upy = (up, up) if isinstance(up, int) else up
    downx, downy = (down, down) if isinstance(down, int) else down
    px0, px1, py0, py1 = _parse_padding(padding)
    n, c, h, w = x.shape
    fh, fw = f.shape
    ow = (w * upx + px0 + px1 - fw + downx) // downx
    oh = (h * upy + py0 + py1 - fh + downy) // downy
    x = x.contiguous()
    f = f.contiguous()
    y = torch.empty([n, c, max(oh, 0), max(ow, 0)], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_upfirdn2d_f32(_p(x), _p(f), _p(y), n * c, h, w, fh, fw, upx, upy, downx, downy,
                                               px0, px1, py0, py1, int(bool(flip_filter)), float(gain), _stream(x)),
                   "upfirdn2d")
    return y


def pack_conv_weight(weight: torch.Tensor):
    """[O,I,3,3] -> (wpk [ceil8(I), 9, ceil32(O)] zero padded, wsq [I,O]) in the kernels' layout
    (same as the host helper nb_pack_conv_weight)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32)
    ip, op = (i + 7) // 8 * 8, (o + 31) // 32 * 32
    wpk = torch.zeros([ip, 9, op], dtype=torch.float32, device=w.device)
    wpk[:i, :, :o] = w.permute(1, 2, 3, 0).reshape(i, 9, o)
    wsq = w.square().sum(dim=[2, 3]).t().contiguous()
    return wpk, wsq


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, *, bias=None, act_gain=None, act_clamp=None,
                     fuse_bias_act=False, x2=None, wpk=None, dcoefs=None):
    """3x3 modulated convolution, reference signature ``networks.modulated_conv2d`` (networks.py:30-88).

    Only the generator's two configurations exist: (up=1, padding=1, flip_weight=True) and
    (up=2, padding=1, flip_weight=False, resample_filter=[1,3,3,1] outer/64).  ``fused_modconv`` is
    accepted for API compatibility; both settings are the same arithmetic here (module docstring of
    csrc/nb_modconv.hip).  Keyword-only extras expose the kernel's fused epilogue
    (``bias``/``act_gain``/``act_clamp`` = the bias_act that follows in SynthesisLayer.forward,
    networks.py:388-390) and the second input range ``x2`` (geometry channels).
    """
    _dev(x, "x"); _dev(weight, "weight"); _dev(styles, "styles")
    n, c1 = x.shape[0], x.shape[1]
    o, i, kh, kw = weight.shape
    c2 = 0 if x2 is None else x2.shape[1]
    assert kh == 3 and kw == 3, "only 3x3 kernels are on the generator path (ToRGB is nb_torgb_triad)"
    assert i == c1 + c2 and styles.shape == (n, i), "shape mismatch"   # misc.assert_shape, networks.py:46-48
    assert down == 1 and padding == 1
    assert (up == 1 and flip_weight) or (up == 2 and not flip_weight), "unsupported up/flip_weight combination"
    h, w_ = x.shape[2], x.shape[3]
    if wpk is None:
        wpk, wsq = pack_conv_weight(weight)
    else:
        wsq = None
    styles = styles.contiguous()
    if dcoefs is None:
        if demodulate:
            if wsq is None:
                wsq = weight.square().sum(dim=[2, 3]).t().contiguous()
            dcoefs = torch.empty([n, o], dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().nb_demod_coefs_f32(_p(styles), _p(wsq), _p(dcoefs), n, i, o, _stream(x)),
                           "demod_coefs")
        else:
            dcoefs = torch.ones([n, o], dtype=torch.float32, device=x.device)
    ho, wo = h * up, w_ * up
    noise_stride = 0
    if noise is not None:
        _dev(noise, "noise")
        noise = noise.contiguous()
        assert noise.numel() in (ho * wo, n * ho * wo)
        noise_stride = ho * wo if noise.numel() == n * ho * wo and n > 1 else 0
    if fuse_bias_act:
        b = bias.contiguous()
        alpha, gain, clamp = 0.2, float(math.sqrt(2) if act_gain is None else act_gain), float(-1 if act_clamp is None else act_clamp)
    else:
        b = torch.zeros([o], dtype=torch.float32, device=x.device)
        alpha, gain, clamp = 1.0, 1.0, -1.0     # identity epilogue
    x = x.contiguous()
    x2c = None if x2 is None else x2.contiguous()
    y = torch.empty([n, o, ho, wo], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_modconv3x3_f32(_p(x), c1, _p(x2c), c2, _p(wpk), _p(styles), _p(dcoefs.contiguous()),
                                                _p(noise), noise_stride, _p(b), _p(y), n, h, w_, o, up, alpha, gain,
                                                clamp, _stream(x)), "modulated_conv2d")
    return y


def blend(features, alpha, x):
    """``BlendedFeatures.blend`` (forger/train/stitching.py:24-25) on the device."""
    _dev(x, "x"); _dev(features, "features"); _dev(alpha, "alpha")
    n, c, h, w = x.shape
    assert features.shape[1:] == (c, h, w) and features.shape[0] in (1, n)
    assert alpha.shape[-2:] == (h, w) and alpha.numel() in (h * w, n * h * w)
    x = x.contiguous()
    y = torch.empty_like(x)
    na = n if alpha.numel() == n * h * w and n > 1 else 1
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_blend_f32(_p(features.contiguous()), features.shape[0], _p(alpha.contiguous()), na,
                                           _p(x), _p(y), n, c, h * w, _stream(x)), "blend")
    return y


# ----------------------------------------------------------------------------------------------
# split-f16 ("h3") fast path (csrc/nb_modconv_h3.hip)
# ----------------------------------------------------------------------------------------------

def pack_conv_weight_h3(weight: torch.Tensor) -> torch.Tensor:
    """[O,I,3,3] fp32 -> static hi/lo f16 weights [ceil(I/16), 3, 3, 2(cg), 2(hi,lo), ceil64(O), 8] (device)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32)
    nch, op = (i + 15) // 16, (o + 63) // 64 * 64
    wp = torch.zeros([nch * 16, 3, 3, op], dtype=torch.float32, device=w.device)
    wp[:i, :, :, :o] = w.permute(1, 2, 3, 0)
    hi = wp.to(torch.float16)
    lo = (wp - hi.to(torch.float32)).to(torch.float16)
    both = torch.stack([hi, lo], dim=0)                                   # [hl, c, ky, kx, o]
    both = both.reshape(2, nch, 2, 8, 3, 3, op)                           # [hl, chunk, cg, j, ky, kx, o]
    return both.permute(1, 4, 5, 2, 0, 6, 3).contiguous()                 # [chunk, ky, kx, cg, hl, o, j]


def pack_conv_weight_h3f8(weight: torch.Tensor) -> torch.Tensor:
    """[O,I,3,3] fp32 -> the "f8" weight format: container of pack_conv_weight_h3 with the lo slots holding
    fp8 e4m3 (w) [cg 0] and fp8((w - f16(w)) * 2^11) [cg 1] of the chunk's 16 channels (include/neube_hip.h)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32)
    nch, op = (i + 15) // 16, (o + 63) // 64 * 64
    wp = torch.zeros([nch * 16, 3, 3, op], dtype=torch.float32, device=w.device)
    wp[:i, :, :, :o] = w.permute(1, 2, 3, 0)
    hi = wp.to(torch.float16)
    lo = wp - hi.to(torch.float32)
    f8 = lambda t: t.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    out = torch.empty([nch, 3, 3, 2, 2, op, 16], dtype=torch.uint8, device=w.device)
    hi_b = hi.reshape(nch, 2, 8, 3, 3, op).permute(0, 3, 4, 1, 5, 2).contiguous().view(torch.uint8)      # [chunk,ky,kx,cg,o,16]
    out[:, :, :, :, 0] = hi_b.reshape(nch, 3, 3, 2, op, 16)
    out[:, :, :, 0, 1] = f8(wp).reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)
    out[:, :, :, 1, 1] = f8(lo * 2048.0).reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)
    return out.view(torch.float16).reshape(nch, 3, 3, 2, 2, op, 8)


def pack_h2f8(x, scale=None, x2=None):
    """fp32 NCHW (x ++ x2) * scale[n,c] -> the "f8" activation format (same container as H2)."""
    _dev(x, "x")
    n, c1, h, w = x.shape
    c2 = 0 if x2 is None else x2.shape[1]
    out = torch.empty(h2_shape(n, c1 + c2, h, w), dtype=torch.float16, device=x.device)
    sc = None if scale is None else scale.contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_pack_h2f8_f32(_p(x.contiguous()), c1, _p(None if x2 is None else x2.contiguous()), c2,
                                               _p(sc), _p(out), n, h * w, _stream(x)), "pack_h2f8")
    return out


def h2_shape(n, c, h, w):
    return [n, (c + 7) // 8, 2, h, w, 8]


def pack_h2(x, scale=None, x2=None):
    """fp32 NCHW (optionally the channel concatenation of x and x2) * scale[n,c] -> H2 f16 tensor."""
    _dev(x, "x")
    n, c1, h, w = x.shape
    c2 = 0 if x2 is None else x2.shape[1]
    out = torch.empty(h2_shape(n, c1 + c2, h, w), dtype=torch.float16, device=x.device)
    sc = None if scale is None else scale.contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nb_pack_h2_f32(_p(x.contiguous()), c1, _p(None if x2 is None else x2.contiguous()), c2,
                                             _p(sc), _p(out), n, h * w, _stream(x)), "pack_h2")
    return out


def unpack_h2(xh2, c):
    """H2 -> fp32 NCHW (hi + lo); a test/debug helper built from torch ops."""
    n, c8, _, h, w, _ = xh2.shape
    v = xh2[:, :, 0].to(torch.float32) + xh2[:, :, 1].to(torch.float32)    # [n, c8, h, w, 8]
    return v.permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)[:, :c].contiguous()


def modconv_up1_h3(x_h2, c_in, w_h3, dcoefs, noise, bias, c_out, act_gain=math.sqrt(2), act_clamp=None, alpha=0.2):
    """conv1-type layer on the f16 matrix cores: H2 input (already style-modulated) -> fp32 NCHW output."""
    n, _, _, h, w, _ = x_h2.shape
    y = torch.empty([n, c_out, h, w], dtype=torch.float32, device=x_h2.device)
    ns = 0
    if noise is not None:
        noise = noise.contiguous()
        ns = h * w if noise.numel() == n * h * w and n > 1 else 0
    with torch.cuda.device(x_h2.device):
        _lib.check(_lib.lib().nb_modconv3x3_up1_h3(_p(x_h2), c_in, _p(w_h3), _p(dcoefs.contiguous()), _p(noise), ns,
                                                   _p(bias.contiguous()), _p(y), n, h, w, c_out, alpha, float(act_gain),
                                                   float(-1 if act_clamp is None else act_clamp), _stream(x_h2)),
                   "modconv3x3_up1_h3")
    return y


def modconv_up2_h3(x_h2, c_in, w_h3, dcoefs, noise, bias, c_out, act_gain=math.sqrt(2), act_clamp=None, alpha=0.2):
    """conv0-type (up=2) layer on the f16 matrix cores: H2 input [n,c_in,h,w] -> fp32 NCHW output [n,c_out,2h,2w]."""
    n, _, _, h, w, _ = x_h2.shape
    y = torch.empty([n, c_out, 2 * h, 2 * w], dtype=torch.float32, device=x_h2.device)
    ns = 0
    if noise is not None:
        noise = noise.contiguous()
        ns = 4 * h * w if noise.numel() == n * 4 * h * w and n > 1 else 0
    with torch.cuda.device(x_h2.device):
        _lib.check(_lib.lib().nb_modconv3x3_up2_h3(_p(x_h2), c_in, _p(w_h3), _p(dcoefs.contiguous()), _p(noise), ns,
                                                   _p(bias.contiguous()), _p(y), n, h, w, c_out, alpha, float(act_gain),
                                                   float(-1 if act_clamp is None else act_clamp), _stream(x_h2)),
                   "modconv3x3_up2_h3")
    return y
