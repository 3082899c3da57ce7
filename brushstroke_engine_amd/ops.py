"""Operator layer: the reference's ``torch_utils/ops`` call surface on top of the HIP C ABI.

Same function names, argument meaning and error behaviour as the reference ops
(``bias_act.bias_act``, ``upfirdn2d.upfirdn2d`` / ``setup_filter``, ``networks.modulated_conv2d``),
fp32, device tensors only; ``bias_act`` and ``upfirdn2d`` also carry the reference's gradients (first and second
order, as autograd Functions whose every evaluation is a HIP launch), the rest is forward only.  Every function launches hand-written gfx950 kernels
through :mod:`brushstroke_engine_amd._lib` on torch's current HIP stream; nothing here computes on
the CPU or falls back to torch ops.
"""
from __future__ import annotations

import math
import threading
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

# name -> (cuda_idx, default alpha, default gain, tensors the gradient needs, has a 2nd-order term); bias_act.py:22-32
ACT_SPECS = {
    "linear":   (1, 0.0, 1.0, "", False),
    "relu":     (2, 0.0, math.sqrt(2), "y", False),
    "lrelu":    (3, 0.2, math.sqrt(2), "y", False),
    "tanh":     (4, 0.0, 1.0, "y", True),
    "sigmoid":  (5, 0.0, 1.0, "y", True),
    "elu":      (6, 0.0, 1.0, "y", True),
    "selu":     (7, 0.0, 1.0, "y", True),
    "softplus": (8, 0.0, 1.0, "y", True),
    "swish":    (9, 0.0, math.sqrt(2), "x", True),
}
ACT_CODES = {k: v[0] for k, v in ACT_SPECS.items()}
ACT_DEFAULTS = {k: (v[1], v[2]) for k, v in ACT_SPECS.items()}


def _stream(t: torch.Tensor) -> int:
    return torch._C._cuda_getCurrentRawStream(t.device.index)          # (the handle only: no Stream object per launch)


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOCTX = _NoCtx()


def _on(device: torch.device):
    """Context that makes ``device`` current for a launch -- a no-op object when it already is (the usual case: one process per
    GPU), which saves a device exchange + a context-manager object per launch on the training path's thousands of launches."""
    return _NOCTX if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(x: torch.Tensor, name: str) -> None:
    if not isinstance(x, torch.Tensor):
        raise AssertionError(f"{name} must be a torch.Tensor")
    if x.device.type != "cuda":
        raise RuntimeError(f"{name} must reside on the GPU (got {x.device}); this build has no CPU path")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {x.dtype})")


def _bias_act_launch(x, b, xref, yref, dy, grad, dim, cfg):
    """One launch of the plugin entry ``bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)``
    (bias_act.cpp:32) on contiguous fp32 device tensors."""
    act, alpha, gain, clamp = cfg
    size_b, step_b = (b.shape[0], x.stride(dim)) if b is not None else (0, 1)
    y = torch.empty_like(x)
    with _on(x.device):
        _lib.check(_lib.lib().nb_bias_act_grad_f32(_p(x), _p(b), _p(xref), _p(yref), _p(dy), _p(y), x.numel(), size_b, step_b,
                                                   grad, ACT_CODES[act], alpha, gain, clamp, _stream(x)), "bias_act")
    return y


class _BiasActGrad(torch.autograd.Function):
    """dx = dy * act'(x + b) * gain (clamped region zeroed); differentiable once more
    (``BiasActCudaGrad``, bias_act.py:173-204)."""

    @staticmethod
    def forward(ctx, dy, x, b, y, dim, cfg):
        dy = dy.contiguous()
        ctx.dim, ctx.cfg = dim, cfg
        ctx.save_for_backward(dy if ACT_SPECS[cfg[0]][4] else None, x, b, y)
        return _bias_act_launch(dy, b, x, y, None, 1, dim, cfg)

    @staticmethod
    def backward(ctx, d_dx):
        d_dx = d_dx.contiguous()
        dy, x, b, y = ctx.saved_tensors
        dim, cfg = ctx.dim, ctx.cfg
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, cfg)
        if ACT_SPECS[cfg[0]][4] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _bias_act_launch(d_dx, b, x, y, dy, 2, dim, cfg)
            if ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
        return d_dy, d_x, d_b, None, None, None


class _BiasAct(torch.autograd.Function):
    """``BiasActCuda`` (bias_act.py:145-171): saves x/b only when the gradient formula needs them."""

    @staticmethod
    def forward(ctx, x, b, dim, cfg):
        y = _bias_act_launch(x, b, None, None, None, 0, dim, cfg)
        ref, second = ACT_SPECS[cfg[0]][3], ACT_SPECS[cfg[0]][4]
        keep_x = "x" in ref or second
        ctx.dim, ctx.cfg = dim, cfg
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if "y" in ref or cfg[3] >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dim, (act, alpha, gain, clamp) = ctx.dim, ctx.cfg
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act != "linear" or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, dim, ctx.cfg)
        if ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None, None


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """Fused bias + activation + gain + clamp with gradients of first and second order.
    Mirrors ``bias_act.bias_act`` (bias_act.py:55-89) and its autograd functions (:145-204): every
    evaluation - forward, d/dx, d2/dx2 - is one launch of the HIP kernel."""
    _dev(x, "x")
    if act not in ACT_CODES:
        raise AssertionError(f"unknown activation {act!r}")
    assert clamp is None or clamp >= 0
    d_alpha, d_gain = ACT_DEFAULTS[act]
    cfg = (act, float(d_alpha if alpha is None else alpha), float(d_gain if gain is None else gain),
           float(-1 if clamp is None else clamp))
    x = x.contiguous()
    if b is not None:
        _dev(b, "b")
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        b = b.contiguous()
    if torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad)):
        return _BiasAct.apply(x, b, dim, cfg)
    return _bias_act_launch(x, b, None, None, None, 0, dim, cfg)


def setup_filter(f: Sequence[float] = (1, 3, 3, 1), device=None, normalize=True, flip_filter=False, gain=1, separable=None):
    """``upfirdn2d.setup_filter`` (upfirdn2d.py:72-116): tap lists shorter than 8 become a 2-D outer product,
    longer ones stay 1-D (separable) unless ``separable`` says otherwise."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
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


def _parse_scaling(s):
    if isinstance(s, int):
        s = [s, s]
    assert isinstance(s, (list, tuple)) and all(isinstance(v, int) for v in s)
    sx, sy = s
    assert sx >= 1 and sy >= 1
    return sx, sy


def _upfirdn2d_launch(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    """One launch of the plugin entry ``upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
    flip, gain)`` (upfirdn2d.cpp:16) on a contiguous NCHW fp32 tensor."""
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    ow = (w * upx + px0 + px1 - fw + downx) // downx
    oh = (h * upy + py0 + py1 - fh + downy) // downy
    x = x.contiguous()
    f2d = f2d.contiguous()
    y = torch.empty([n, c, max(oh, 0), max(ow, 0)], dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.lib().nb_upfirdn2d_f32(_p(x), _p(f2d), _p(y), n * c, h, w, fh, fw, upx, upy, downx, downy,
                                               px0, px1, py0, py1, int(bool(flip)), float(gain), _stream(x)),
                   "upfirdn2d")
    return y


def _upfirdn2d_apply(x, f, cfg):
    upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = cfg
    if f.ndim == 2:
        return _upfirdn2d_launch(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
    # separable taps: a row pass then a column pass, sqrt(gain) each (upfirdn2d.py:234-236)
    y = _upfirdn2d_launch(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip, math.sqrt(gain))
    return _upfirdn2d_launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip, math.sqrt(gain))


class _Upfirdn2d(torch.autograd.Function):
    """``Upfirdn2dCuda`` (upfirdn2d.py:223-264): the gradient w.r.t. x is another upfirdn2d with up and down
    swapped, the filter flipped the other way and padding that maps the output grid back onto the input;
    differentiable to any order because the backward is the same Function.  No gradient w.r.t. f."""

    @staticmethod
    def forward(ctx, x, f, cfg):
        ctx.cfg, ctx.x_shape = cfg, x.shape
        ctx.save_for_backward(f)
        return _upfirdn2d_apply(x, f, cfg)

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = ctx.cfg
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = dy.shape
        fw, fh = f.shape[-1], f.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            back = (downx, downy, upx, upy,
                    fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
                    fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1, not flip, gain)
            dx = _Upfirdn2d.apply(dy.contiguous(), f, back)
        assert not ctx.needs_input_grad[1], "upfirdn2d has no gradient with respect to the filter"
        return dx, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Pad, upsample, FIR-filter, downsample, with gradients of any order w.r.t. x.  Mirrors
    ``upfirdn2d.upfirdn2d`` (upfirdn2d.py:120-164): ``f`` is [fh, fw] (non-separable), [taps] (separable) or None."""
    _dev(x, "x")
    assert x.ndim == 4
    if f is None:
        f = _const(1.0, [1, 1], x.device)
    _dev(f, "f")
    assert f.ndim in (1, 2), "f must be rank 1 or 2"
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    cfg = (upx, upy, downx, downy, px0, px1, py0, py1, bool(flip_filter), float(gain))
    if torch.is_grad_enabled() and x.requires_grad:
        return _Upfirdn2d.apply(x, f, cfg)
    return _upfirdn2d_apply(x, f, cfg)


def _filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    """``upfirdn2d.filter2d`` (upfirdn2d.py:268-301): FIR filter, output padded to the input's shape."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """``upfirdn2d.upsample2d`` (upfirdn2d.py:305-340): output is ``up`` times the input, gain scaled by up_x*up_y."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """``upfirdn2d.downsample2d`` (upfirdn2d.py:344-379): output is the input divided by ``down``."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def pack_conv_weight(weight: torch.Tensor, want_wsq: bool = True):
    """[O,I,3,3] -> (wpk [ceil8(I), 9, ceil32(O)] zero padded, wsq [I,O]) in the kernels' layout
    (same as the host helper nb_pack_conv_weight).  ``want_wsq=False``: wsq is None (callers that bring their own
    demodulation coefficients)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32)
    ip, op = (i + 7) // 8 * 8, (o + 31) // 32 * 32
    if ip == i and op == o:
        wpk = w.permute(1, 2, 3, 0).reshape(i, 9, o).contiguous()          # (no padding: one copy)
    else:
        wpk = torch.zeros([ip, 9, op], dtype=torch.float32, device=w.device)
        wpk[:i, :, :o] = w.permute(1, 2, 3, 0).reshape(i, 9, o)
    wsq = w.square().sum(dim=[2, 3]).t().contiguous() if want_wsq else None
    return wpk, wsq


_CONSTS = {}


def _const(value: float, shape, device) -> torch.Tensor:
    """Read-only constant tensors (all-zeros bias, all-ones scales) of the operator wrappers, created once per shape."""
    key = (float(value), tuple(shape), device.type, device.index)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full(list(shape), float(value), dtype=torch.float32, device=device)
    return t


def _pow2(v: int) -> bool:
    return v >= 4 and (v & (v - 1)) == 0


def _conv2d_launch(x, w, in_scale, out_scale, stride, padding):
    n, ci, h, wd = x.shape
    co, ci2, kh, kw = w.shape
    assert ci == ci2
    if kh == 3 and kw == 3 and stride == 1 and padding == 1 and _pow2(h) and _pow2(wd):
        # the tiled fp32-MFMA kernel of the generator (LDS-staged, 5-20x the generic kernel's rate): its styles / demodulation
        # slots carry the optional channel scales (ones otherwise), identity epilogue
        ones = lambda c: _const(1.0, [n, c], x.device)
        return _modulated_conv2d_forward(x, w, ones(ci) if in_scale is None else in_scale, None, up=1, padding=1, demodulate=False,
                                         flip_weight=True, dcoefs=ones(co) if out_scale is None else out_scale)
    if _s2_valid_h3_eligible(n, ci, h, wd, kh, kw, stride, padding):
        return _conv2d_s2_valid_h3(x, w, in_scale, out_scale)
    ho, wo = (h + 2 * padding - kh) // stride + 1, (wd + 2 * padding - kw) // stride + 1
    if (TRAIN_SPLIT_F16 and kh == 3 and kw == 3 and stride == 1 and padding in (0, 1, 2) and ho >= 16 and wo >= 16
            and n * ho * wo >= TRAIN_SPLIT_F16_MIN_PIXELS):
        # odd-sized stride-1 correlations (the zero-stuffed / (2H+1)-sized grids of the create-graph passes and of transposed
        # convolutions): embed the image -- shifted by padding - 1, so that a "same" convolution reproduces this padding -- in
        # zeros of a size the tiled kernels take, run those, and cut the result out.  <= 1.4x the multiply-adds for 10-20x the rate
        # of the generic kernel.
        sh, off = max(padding - 1, 0), (1 if padding == 0 else 0)          # image shift inside the canvas, offset of the result
        hp, wp = (max(ho + off, h + sh) + 15) // 16 * 16, (max(wo + off, wd + sh) + 31) // 32 * 32
        xp = torch.nn.functional.pad(x, (sh, wp - wd - sh, sh, hp - h - sh))
        ones = lambda c: _const(1.0, [n, c], x.device)
        yp = _modulated_conv2d_forward(xp.contiguous(), w, ones(ci) if in_scale is None else in_scale, None, up=1, padding=1,
                                       demodulate=False, flip_weight=True, dcoefs=ones(co) if out_scale is None else out_scale)
        return yp[:, :, off:off + ho, off:off + wo].contiguous()
    y = torch.empty([n, co, ho, wo], dtype=torch.float32, device=x.device)
    isc = None if in_scale is None else in_scale.contiguous()
    osc = None if out_scale is None else out_scale.contiguous()
    with _on(x.device):
        _lib.check(_lib.lib().nb_conv2d_f32(_p(x.contiguous()), _p(w.contiguous()), _p(isc), _p(osc), _p(y), n, ci, h, wd, co, kh, kw,
                                            stride, padding, _stream(x)), "conv2d")
    return y


# Weight-gradient correlations on the f16 matrix cores (split operands with power-of-two range scaling, pipelined staging:
# nb_conv2d_wgrad_h3, 1.8x the exact-fp32 MFMA kernel on the 128-channel 128x128 layers); False: the exact-fp32 kernel.
WGRAD_SPLIT_F16 = True


class _RangeSlots:
    """Zero-filled 4-byte device words for nb_absmax_f32, handed out pairwise from a ring that is re-zeroed with ONE fill per
    lap (stream order keeps a pair's consumers ahead of the fill that recycles it: everything here runs on the current stream)."""
    SIZE = 8192
    _rings = {}
    _lock = threading.Lock()                                            # (forward and autograd threads both come here)

    @classmethod
    def pair(cls, device) -> torch.Tensor:
        key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
        with cls._lock:
            ring = cls._rings.get(key)
            if ring is None or ring[1] + 4 > cls.SIZE:
                with _on(device):
                    buf = torch.zeros([cls.SIZE], dtype=torch.int32, device=device) if ring is None else ring[0].zero_()
                ring = cls._rings[key] = [buf, 0]
            out = ring[0][ring[1]:ring[1] + 2]
            ring[1] += 4                                                # 16-byte pitch
        return out


def _absmax_slots(a, b=None, c=None) -> torch.Tensor:
    """Two device words: max(|a|, |b|) and max|c| as float bit patterns, in one launch (no host sync)."""
    slots = _RangeSlots.pair(a.device)
    ts = [None if t is None else (t if t.is_contiguous() else t.contiguous()) for t in (a, b, c)]      # (pointers only: no detach)
    with _on(a.device):
        _lib.check(_lib.lib().nb_absmax_f32(_p(ts[0]), ts[0].numel(), _p(ts[1]), 0 if ts[1] is None else ts[1].numel(),
                                            _p(ts[2]), 0 if ts[2] is None else ts[2].numel(), _p(slots), _stream(a)), "absmax")
    return slots


def _wgrad_launch(u, v, stride, padding, sum_n=False):
    n, cu, hu, wu = u.shape
    n2, cv, hv, wv = v.shape
    assert n == n2
    if WGRAD_SPLIT_F16:
        # each operand is brought near 2^10 by a power of two the kernel derives from its max-abs slot (no sync, 3 launches)
        u, v = u.contiguous(), v.contiguous()
        slots = _absmax_slots(u, None, v)
        a = torch.empty(([] if sum_n else [n]) + [cu, cv, 3, 3], dtype=torch.float32, device=u.device)
        L = _lib.lib()
        nbytes = int(L.nb_conv2d_wgrad_h3_ws_bytes(n, cu, cv, hv, int(sum_n)))
        ws = torch.empty([nbytes // 4], dtype=torch.float32, device=u.device) if nbytes else None
        with _on(u.device):
            _lib.check(L.nb_conv2d_wgrad_h3_ws(_p(u), _p(v), _p(slots), 1, _p(a), _p(ws), nbytes, int(sum_n),
                                               n, cu, hu, wu, cv, hv, wv, stride, padding, _stream(u)), "conv2d_wgrad_h3")
        return a
    a = torch.empty([n, cu, cv, 3, 3], dtype=torch.float32, device=u.device)
    with _on(u.device):
        _lib.check(_lib.lib().nb_conv2d_wgrad_f32(_p(u.contiguous()), _p(v.contiguous()), _p(a), n, cu, hu, wu, cv, hv, wv,
                                                  stride, padding, _stream(u)), "conv2d_wgrad")
    return a.sum(dim=0) if sum_n else a


def _conv2d_input_grad(dy, w, x_shape, stride, padding):
    """Gradient of conv2d w.r.t. its input = transposed convolution, built from differentiable pieces: zero-stuffing by
    ``stride`` (upfirdn2d with the identity filter), a stride-1 correlation with the transposed, flipped kernel, and zero
    rows / columns where the forward window never reached."""
    kh, kw = w.shape[2], w.shape[3]
    h, wd = x_shape[2], x_shape[3]
    if (not torch.is_grad_enabled() and stride == 1 and kh == 3 and kw == 3 and padding == 1 and _pow2(h) and _pow2(wd)
            and tuple(dy.shape[2:]) == (h, wd)):
        # first-order pass: the tiled kernels take the forward weight and pack its transposed, tap-reversed form themselves
        n = dy.shape[0]
        return _modulated_conv2d_forward(dy.contiguous(), w.detach(), _const(1.0, [n, w.shape[0]], dy.device), None, up=1, padding=1,
                                         demodulate=False, flip_weight=True, dcoefs=_const(1.0, [n, w.shape[1]], dy.device), weight_tf=True)
    if stride > 1:
        dy = upfirdn2d(dy, None, up=stride, padding=[0, -(stride - 1), 0, -(stride - 1)])
    assert kh - 1 - padding >= 0 and kw - 1 - padding >= 0, "conv2d gradient: padding larger than kernel - 1"
    wt = w.transpose(0, 1).flip([2, 3])
    if kh != kw:
        raise NotImplementedError("conv2d gradient: square kernels only")
    dx = conv2d(dy, wt, stride=1, padding=kh - 1 - padding)
    ph, pw = h - dx.shape[2], wd - dx.shape[3]
    if ph or pw:
        dx = torch.nn.functional.pad(dx, (0, pw, 0, ph))
    return dx


class _Conv2d(torch.autograd.Function):
    """Plain convolution with gradients of any order (``conv2d_gradfix.py:107-168``): every gradient is expressed through
    this Function and :class:`_Conv2dWgrad` again."""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.stride, ctx.padding = stride, padding
        ctx.save_for_backward(x, w)
        return _conv2d_launch(x, w, None, None, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _conv2d_input_grad(dy, w, x.shape, ctx.stride, ctx.padding)
        if ctx.needs_input_grad[1]:
            dw = _Conv2dWgrad.apply(x, dy, ctx.stride, ctx.padding, w.shape[2])
        return dx, dw, None, None


class _Conv2dWgrad(torch.autograd.Function):
    """dw[co,ci,a,b] = sum_{n,i,j} x[n,ci,i*stride+a-pad,j*stride+b-pad] * dy[n,co,i,j] for 3x3 and 1x1 kernels (the 1x1
    case is the centre tap of the 3x3 correlation); differentiable (R1-type double backward)."""

    @staticmethod
    def forward(ctx, x, dy, stride, padding, k):
        assert k in (1, 3), "conv2d weight gradient: 1x1 and 3x3 kernels"
        ctx.stride, ctx.padding, ctx.k = stride, padding, k
        ctx.save_for_backward(x, dy)
        a = _wgrad_launch(x, dy, stride, padding + (1 if k == 1 else 0), sum_n=True)         # [ci, co, 3, 3]
        a = a.permute(1, 0, 2, 3)
        return a[:, :, 1:2, 1:2].contiguous() if k == 1 else a.contiguous()

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        dx = ddy = None
        if ctx.needs_input_grad[0]:
            dx = _conv2d_input_grad(dy, ddw, x.shape, ctx.stride, ctx.padding)
        if ctx.needs_input_grad[1]:
            ddy = conv2d(x, ddw, stride=ctx.stride, padding=ctx.padding)
        return dx, ddy, None, None, None


def conv2d(x, w, in_scale=None, out_scale=None, stride=1, padding=0):
    """Generic fp32 cross-correlation with zero padding on the HIP kernel (nb_conv2d_f32) - the role cuDNN plays behind
    ``conv2d_gradfix``.  Without channel scales it is differentiable to any order w.r.t. x and w (1x1 / 3x3 kernels for
    the weight gradient); the per-sample ``in_scale`` / ``out_scale`` form is forward only."""
    _dev(x, "x"); _dev(w, "w")
    if in_scale is None and out_scale is None and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _Conv2d.apply(x, w, int(stride), int(padding))
    return _conv2d_launch(x, w, in_scale, out_scale, int(stride), int(padding))


def conv2d_wgrad(u, v, stride=1, padding=0):
    """a[n,cu,cv,ka,kb] = sum_{i,j} u[n,cu,i*stride+ka-pad,j*stride+kb-pad] * v[n,cv,i,j]  (3x3; nb_conv2d_wgrad_f32)."""
    _dev(u, "u"); _dev(v, "v")
    return _wgrad_launch(u, v, stride, padding)


_FIR1331 = None


def _is_fir1331(f: torch.Tensor) -> bool:
    """f is the [1,3,3,1] x [1,3,3,1] / 64 filter the fused up=2 kernels have built in (checked once per tensor)."""
    if f is None or f.ndim != 2 or tuple(f.shape) != (4, 4):
        return False
    tag = getattr(f, "_nb_is_fir1331", None)
    if tag is None or tag[0] != f._version:
        global _FIR1331
        if _FIR1331 is None or _FIR1331.device != f.device:
            t = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float32, device=f.device)
            _FIR1331 = torch.outer(t, t) / 64
        tag = (f._version, bool(torch.equal(f.detach().to(torch.float32), _FIR1331)))
        f._nb_is_fir1331 = tag
    return tag[1]


class _Down2Conv2d(torch.autograd.Function):
    """``conv2d_resample(x, w, f, down=2, padding=1)`` for a 3x3 kernel (conv2d_resample.py:96-113: FIR with padding 2, then the
    stride-2 correlation) -- the discriminator's down-sampling convolution -- as one differentiable operator, so that its input
    gradient does not have to be assembled from zero-stuffing + a stride-1 correlation over a 3/4-empty image + the FIR adjoint:

      dx = FIR_adj(convT_{s2}(dy, w)) = 1/4 * [up=2 layer of the generator](dy, weight = w^T)

    (the transposed convolution followed by the same symmetric FIR, up to the gain), i.e. ONE launch of the fused 4-phase up=2
    kernel; dw is the stride-2 weight-gradient correlation of the filtered input with dy.  Under create_graph (R1) the gradient
    is rebuilt from the generic differentiable operators."""

    @staticmethod
    def forward(ctx, x, w, f):
        t = _upfirdn2d_apply(x.contiguous(), f, (1, 1, 1, 1, 2, 2, 2, 2, False, 1.0))
        y = _conv2d_launch(t, w, None, None, 2, 0)
        ctx.save_for_backward(x, t, w, f)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, t, w, f = ctx.saved_tensors
        if torch.is_grad_enabled():
            inputs = [v for v, need in zip((x, w), ctx.needs_input_grad[:2]) if need]
            with torch.enable_grad():
                y2 = conv2d(upfirdn2d(x, f, padding=[2, 2, 2, 2]), w, stride=2, padding=0)
                grads = list(torch.autograd.grad(y2, inputs, dy, create_graph=True, allow_unused=True))
            return tuple(grads.pop(0) if need else None for need in ctx.needs_input_grad[:2]) + (None,)
        dy = dy.contiguous()
        dx = dw = None
        n, co = dy.shape[0], dy.shape[1]
        if ctx.needs_input_grad[0]:
            wt = w.detach().transpose(0, 1).contiguous()                     # [ci, co, 3, 3]: the up=2 layer's weight
            dx = _modulated_conv2d_forward(dy, wt, _const(1.0, [n, co], dy.device), None, up=2, padding=1, demodulate=False,
                                           flip_weight=False, dcoefs=_const(0.25, [n, wt.shape[0]], dy.device))
        if ctx.needs_input_grad[1]:
            dw = _wgrad_launch(t, dy, 2, 0, sum_n=True).permute(1, 0, 2, 3).contiguous()
        return dx, dw, None


def conv2d_down2(x, w, f):
    """FIR (padding 2) + stride-2 3x3 correlation = ``conv2d_resample(x, w, f, down=2, padding=1)`` (conv2d_resample.py:96-113)."""
    _dev(x, "x"); _dev(w, "w")
    h, wd = x.shape[2], x.shape[3]
    fused = (tuple(w.shape[2:]) == (3, 3) and _is_fir1331(f) and h == wd and _pow2(h) and h >= 16)
    if fused and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _Down2Conv2d.apply(x, w, f)
    return conv2d(upfirdn2d(x, f, padding=[2, 2, 2, 2]), w, stride=2, padding=0)


class _ModulatedConv2d(torch.autograd.Function):
    """First-order gradients of the fused modulated convolution (what autograd + ``conv2d_gradfix`` + the upfirdn2d
    backward give the reference, networks.py:30-88).  With xs = x * s, z = conv_resample(xs, W), y = z * d + noise,
    d = rsqrt(sum (W s)^2 + 1e-8):

      dz = dy * d;  dd = sum_pix dy * z;  dq = -1/2 d^3 dd   (q = the sum under the rsqrt)
      up = 1:  dx = s * corr(dz, W^T flipped)  - the fused forward kernel with the roles of styles and demodulation
               swapped (x <- dy, styles <- d, weights <- W^T flipped, dcoefs <- s);
               A[n,o,c] = sum_pix dz[n,o,p] x[n,c,p+tap]          (nb_conv2d_wgrad_f32, U = x pad 1, V = dz)
      up = 2:  g1 = FIR-adjoint of dz on the (2H+1)^2 grid (upfirdn2d with the filter flipped the other way),
               dx = s * conv(g1, W^T, stride 2)                    (nb_conv2d_f32);
               A[n,o,c] = sum_{i,j} g1[n,o,2i+a,2j+b] x[n,c,i,j]   (nb_conv2d_wgrad_f32, U = g1, V = x, stride 2)
      dW = sum_n s A + 2 W sum_n dq s^2;   ds = sum_{o,tap} W A + 2 s (dq @ Wsq);   dnoise = dy (summed over broadcast dims)
    """

    @staticmethod
    def forward(ctx, x, weight, styles, noise, up, resample_filter, demodulate):
        n, o = x.shape[0], weight.shape[0]
        wsq = weight.detach().square().sum(dim=[2, 3])                       # [O, C]
        if demodulate:
            d = (styles.detach().square() @ wsq.t() + 1e-8).rsqrt()          # [N, O]  (tiny GEMM: plumbing)
        else:
            d = _const(1.0, [n, o], x.device)
        y = _modulated_conv2d_forward(x.detach(), weight.detach(), styles.detach(), None if noise is None else noise.detach(),
                                      up=up, padding=1, resample_filter=resample_filter, demodulate=demodulate,
                                      flip_weight=(up == 1), dcoefs=d)
        ctx.up, ctx.demodulate = up, demodulate
        ctx.noise_shape = None if noise is None else noise.shape
        ctx.save_for_backward(x, weight, styles, noise, d, y, resample_filter)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, styles, noise, d, y, f = ctx.saved_tensors
        up = ctx.up
        if torch.is_grad_enabled():
            # create_graph=True (path-length regulariser, loss_modified.py:205-221): the gradients themselves must be
            # differentiable.  Re-evaluate the layer through the generic differentiable operators and let autograd
            # produce (and later differentiate) them - slower kernels, used only in this mode.
            inputs = [t for t, need in zip((x, weight, styles, noise), ctx.needs_input_grad[:4]) if need and t is not None]
            with torch.enable_grad():
                y2 = _modulated_conv2d_generic(x, weight, styles, noise, up, f, ctx.demodulate)
                grads = list(torch.autograd.grad(y2, inputs, dy, create_graph=True, allow_unused=True))
            out = []
            for t, need in zip((x, weight, styles, noise), ctx.needs_input_grad[:4]):
                out.append(grads.pop(0) if (need and t is not None) else None)
            return out[0], out[1], out[2], out[3], None, None, None
        dy = dy.contiguous()
        n, c, h, w_ = x.shape
        o = weight.shape[0]
        need_x, need_w, need_s, need_nz = ctx.needs_input_grad[:4]
        dx = dw = ds = dnz = None
        if need_nz and noise is not None:
            dnz = dy
            for dim, (a, b) in enumerate(zip(dy.shape, ctx.noise_shape)):
                if b == 1 and a != 1:
                    dnz = dnz.sum(dim=dim, keepdim=True)
        s = styles.detach()
        wd = weight.detach()
        if up == 1:
            if need_x:
                # the correlation with weight^T, taps reversed ([C, O, 3, 3]): packed straight from ``weight``
                dx = _modulated_conv2d_forward(dy, wd, d, None, up=1, padding=1, demodulate=False, flip_weight=True, dcoefs=s, weight_tf=True)
            if need_w or need_s:
                dz = dy * d[:, :, None, None]
                A = conv2d_wgrad(x.detach(), dz, stride=1, padding=1)        # [N, C, O, 3, 3]
                a_strides = (c * o * 9, 9, o * 9)                            # element strides of (n, o, c)
        else:
            dz = dy * d[:, :, None, None]
            fh, fw = f.shape
            # adjoint of upfirdn2d(y1, f, padding=1, gain=4): pad fw - 1 - 1, filter flipped the other way
            g1 = _upfirdn2d_apply(dz, f, (1, 1, 1, 1, fw - 2, fw - 2, fh - 2, fh - 2, True, 4.0))
            if need_x:
                dx = conv2d(g1, wd.transpose(0, 1), out_scale=s, stride=2, padding=0)
            if need_w or need_s:
                A = conv2d_wgrad(g1, x.detach(), stride=2, padding=0)        # [N, O, C, 3, 3]
                a_strides = (o * c * 9, c * 9, 9)
        if need_w or need_s:
            # the rest in three small launches (nb_modconv_bwd_dot_f32 / _finish_f32): dd = sum_pix dy (y - noise), then
            #   dW = sum_n s A + 2 W sum_n dq s^2,   ds = sum_{o,tap} W A + 2 s (dq @ Wsq),   dq = -1/2 d^3 dd / d
            L = _lib.lib()
            s_c, w_c = s.contiguous(), wd.contiguous()
            dq = None
            with _on(x.device):
                if ctx.demodulate:
                    dd = torch.empty([n, o], dtype=torch.float32, device=x.device)
                    yc = y.contiguous()
                    nz, nz_stride = None, 0
                    if noise is not None:
                        nz = noise.detach().contiguous()
                        assert nz.numel() in (yc.shape[2] * yc.shape[3], n * yc.shape[2] * yc.shape[3]), "noise must be one plane or one per sample"
                        nz_stride = yc.shape[2] * yc.shape[3] if nz.numel() == n * yc.shape[2] * yc.shape[3] and n > 1 else 0
                    _lib.check(L.nb_modconv_bwd_dot_f32(_p(dy), _p(yc), _p(nz), nz_stride, _p(dd), n, o, yc.shape[2] * yc.shape[3], _stream(x)),
                               "modconv_bwd_dot")
                    dq = (-0.5 * d.square() * dd).contiguous()               # -1/2 d^3 (dd / d)
                dw = torch.empty_like(w_c) if need_w else None
                ds = torch.empty_like(s_c) if need_s else None
                _lib.check(L.nb_modconv_bwd_finish_f32(_p(A), a_strides[0], a_strides[1], a_strides[2], _p(s_c), _p(w_c), _p(dq), _p(dw), _p(ds),
                                                       n, o, c, _stream(x)), "modconv_bwd_finish")
        return dx, dw, ds, dnz, None, None, None


def _modulated_conv2d_generic(x, weight, styles, noise, up, f, demodulate):
    """The layer in its non-fused form (networks.py:67-76) on the generic differentiable operators: x*s -> shared conv
    (up = 2: zero-stuffing + correlation with the flipped kernel = the stride-2 transposed conv, then the FIR) -> *d + noise."""
    n = x.shape[0]
    xs = x * styles.reshape(n, -1, 1, 1)
    if up == 1:
        z = conv2d(xs, weight, stride=1, padding=1)
    else:
        stuffed = upfirdn2d(xs, None, up=2, padding=[0, -1, 0, -1])
        y1 = conv2d(stuffed, weight.flip([2, 3]), stride=1, padding=2)
        z = upfirdn2d(y1, f, padding=1, gain=4)
    if demodulate:
        d = ((styles.square() @ weight.square().sum(dim=[2, 3]).t()) + 1e-8).rsqrt()
        z = z * d.reshape(n, -1, 1, 1)
    return z if noise is None else z + noise


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, *, bias=None, act_gain=None, act_clamp=None,
                     fuse_bias_act=False, x2=None, wpk=None, dcoefs=None):
    """3x3 modulated convolution, reference signature ``networks.modulated_conv2d`` (networks.py:30-88), with first-order
    gradients w.r.t. x, weight, styles and noise when any of them requires grad (plain configuration: no fused bias /
    second input) and, under ``create_graph=True``, gradients of those gradients; see :class:`_ModulatedConv2d`."""
    wants_grad = torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad
                                                 for t in (x, weight, styles, noise))
    if wants_grad:
        assert not fuse_bias_act and x2 is None and wpk is None and dcoefs is None and down == 1 and padding == 1, \
            "gradients are implemented for the plain modulated_conv2d configuration"
        assert (up == 1 and flip_weight) or (up == 2 and not flip_weight), "unsupported up/flip_weight combination"
        if up == 2:
            assert resample_filter is not None and resample_filter.ndim == 2
        return _ModulatedConv2d.apply(x, weight, styles, noise, up, resample_filter, demodulate)
    return _modulated_conv2d_forward(x, weight, styles, noise=noise, up=up, down=down, padding=padding,
                                     resample_filter=resample_filter, demodulate=demodulate, flip_weight=flip_weight,
                                     fused_modconv=fused_modconv, bias=bias, act_gain=act_gain, act_clamp=act_clamp,
                                     fuse_bias_act=fuse_bias_act, x2=x2, wpk=wpk, dcoefs=dcoefs)


# The differentiable operators (training path) evaluate their stride-1 / up=2 3x3 convolutions on the split-f16 kernels of the
# inference path when the shapes allow it (forward passes, input gradients, the discriminator's convolutions through
# ops.conv2d): hi/lo-f16 operands = 22-bit products, fp32 accumulation -- the grade of an fp32 evaluation, ~4x the fp32-MFMA rate.
# Operands are brought into the f16 range by a power-of-two scale computed on the device (gradients are tiny, activations can
# be large) and the result is scaled back in the kernel's output coefficients.  False: the exact-fp32 MFMA kernels.
TRAIN_SPLIT_F16 = True
TRAIN_SPLIT_F16_MIN_PIXELS = 8192               # n * h * w (x up^2) from which the split-f16 kernels are used (measured: 32 x 32 at batch 8 already pays)


def _absmax(t: torch.Tensor) -> torch.Tensor:
    """max |t| as a 0-dim device tensor in ONE reduction pass (aminmax), no host sync."""
    lo, hi = torch.aminmax(t.detach())
    return torch.maximum(hi, -lo)


def pack_conv_weight_h3_dev(weight: torch.Tensor, co_align: int = 64, tf: bool = False) -> torch.Tensor:
    """:func:`pack_conv_weight_h3` in one HIP launch (weights that change every step); ``co_align`` = 128 gives the
    encoder-type kernels' container; ``tf``: pack ``weight.transpose(0, 1).flip([2, 3])`` without materialising it."""
    o, i, kh, kw = weight.shape
    if tf:
        o, i = i, o
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32).contiguous()
    nch, op = (i + 15) // 16, (o + co_align - 1) // co_align * co_align
    out = torch.empty([nch, 3, 3, 2, 2, op, 8], dtype=torch.float16, device=w.device)
    with _on(w.device):
        _lib.check(_lib.lib().nb_pack_conv_weight_h3_dev(_p(w), o, i, co_align, int(tf), _p(out), _stream(w)), "pack_conv_weight_h3_dev")
    return out


def _s2_valid_h3_eligible(n, ci, h, wd, kh, kw, stride, padding) -> bool:
    if not (TRAIN_SPLIT_F16 and kh == 3 and kw == 3 and stride == 2 and padding == 0 and h % 2 == 1 and wd % 2 == 1):
        return False
    ho, wo = (h - 1) // 2, (wd - 1) // 2
    return n * ho * wo >= 4096 and ((wo % 32 == 0 and ho % 8 == 0) or (wo == 16 and ho % 16 == 0))


def _conv2d_s2_valid_h3(x, w, in_scale, out_scale):
    """Stride-2 3x3 correlation without padding on the split-f16 kernel (nb_conv3x3_s2_valid_h3): operands range-scaled and
    packed on the device as in the modulated convolutions; 4 launches."""
    n, ci, h, wd = x.shape
    co = w.shape[0]
    ho, wo = (h - 1) // 2, (wd - 1) // 2
    x = x.contiguous()
    isc = _const(1.0, [n, ci], x.device) if in_scale is None else in_scale.contiguous()
    slots = _absmax_slots(x, None, None if in_scale is None else isc)
    dco_in = _const(1.0, [n, co], x.device) if out_scale is None else out_scale.contiguous()
    dco = torch.empty_like(dco_in)
    xh = torch.empty(h2_shape(n, ci, h, wd), dtype=torch.float16, device=x.device)
    y = torch.empty([n, co, ho, wo], dtype=torch.float32, device=x.device)
    L = _lib.lib()
    with _on(x.device):
        _lib.check(L.nb_pack_h2_ranged_f32(_p(x), ci, None, 0, _p(isc), _p(xh), n, h * wd, _p(slots), 16384.0, _p(dco_in), _p(dco),
                                           dco_in.numel(), _stream(x)), "pack_h2_ranged")
        wh = pack_conv_weight_h3_dev(w, co_align=128)
        _lib.check(L.nb_conv3x3_s2_valid_h3(_p(xh), ci, _p(wh), _p(_const(0.0, [co], x.device)), _p(dco), co, _p(y), n, h, wd, co,
                                            _stream(x)), "conv3x3_s2_valid_h3")
    return y


def _split_f16_eligible(n, h, w_, up) -> bool:
    if not TRAIN_SPLIT_F16 or n * h * w_ * up * up < TRAIN_SPLIT_F16_MIN_PIXELS:
        return False
    if up == 1:
        return w_ % 32 == 0 and h % 16 == 0
    return (w_ % 32 == 0 or w_ == 16) and h >= 8


def _modulated_conv2d_forward(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                              flip_weight=True, fused_modconv=True, *, bias=None, act_gain=None, act_clamp=None,
                              fuse_bias_act=False, x2=None, wpk=None, dcoefs=None, weight_tf=False):
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
    h, w_ = x.shape[2], x.shape[3]
    if weight_tf and not (wpk is None and _split_f16_eligible(n, h, w_, up)):
        # (weight_tf: ``weight`` stands for weight.transpose(0, 1).flip([2, 3]) -- only the device pack of the split-f16 path
        #  takes it as it is)
        weight, weight_tf = weight.transpose(0, 1).flip([2, 3]).contiguous(), False
    o, i, kh, kw = weight.shape
    if weight_tf:
        o, i = i, o
    c2 = 0 if x2 is None else x2.shape[1]
    assert kh == 3 and kw == 3, "only 3x3 kernels are on the generator path (ToRGB is nb_torgb_triad)"
    assert i == c1 + c2 and styles.shape == (n, i), "shape mismatch"   # misc.assert_shape, networks.py:46-48
    assert down == 1 and padding == 1
    assert (up == 1 and flip_weight) or (up == 2 and not flip_weight), "unsupported up/flip_weight combination"
    h, w_ = x.shape[2], x.shape[3]
    wpk_given = wpk
    use_h3 = wpk is None and _split_f16_eligible(n, h, w_, up)
    if wpk is None:
        wpk, wsq = (None, None) if use_h3 else pack_conv_weight(weight, want_wsq=(dcoefs is None and demodulate))
    else:
        wsq = None
    styles = styles.contiguous()
    if dcoefs is None:
        if demodulate:
            if wsq is None:
                wsq = weight.square().sum(dim=[2, 3]).t().contiguous()
            dcoefs = torch.empty([n, o], dtype=torch.float32, device=x.device)
            with _on(x.device):
                _lib.check(_lib.lib().nb_demod_coefs_f32(_p(styles), _p(wsq), _p(dcoefs), n, i, o, _stream(x)),
                           "demod_coefs")
        else:
            dcoefs = _const(1.0, [n, o], x.device)
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
        b = _const(0.0, [o], x.device)
        alpha, gain, clamp = 1.0, 1.0, -1.0     # identity epilogue
    x = x.contiguous()
    x2c = None if x2 is None else x2.contiguous()
    if wpk_given is None and _split_f16_eligible(n, h, w_, up):
        # split-f16 kernels: (x ++ x2) * styles * 2^k -> H2 operands (k from the operands' max-abs, on the device), weights
        # packed on the device, 2^-k folded into the output coefficients: 5 launches
        slots = _absmax_slots(x, x2c, styles)
        dco_in = dcoefs.contiguous()
        dco = torch.empty_like(dco_in)
        xh = torch.empty(h2_shape(n, i, h, w_), dtype=torch.float16, device=x.device)
        with _on(x.device):
            _lib.check(_lib.lib().nb_pack_h2_ranged_f32(_p(x), c1, _p(x2c), c2, _p(styles), _p(xh), n, h * w_, _p(slots), 16384.0,
                                                        _p(dco_in), _p(dco), dco_in.numel(), _stream(x)), "pack_h2_ranged")
        wh = pack_conv_weight_h3_dev(weight, tf=weight_tf)
        fn = modconv_up1_h3 if up == 1 else modconv_up2_h3
        return fn(xh, i, wh, dco, noise, b, o, act_gain=gain, act_clamp=None if clamp < 0 else clamp, alpha=alpha)
    y = torch.empty([n, o, ho, wo], dtype=torch.float32, device=x.device)
    with _on(x.device):
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
    with _on(x.device):
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


def fold_up2_fir(weight: torch.Tensor, resample_filter: torch.Tensor) -> torch.Tensor:
    """Effective per-phase 3x3 kernels of `stride-2 transposed conv + 4x4 FIR (pad 1, gain 4)` (conv2d_resample.py:124-142,
    upfirdn2d.py:168-208): out[2m+py, 2n+px] = sum_{di,dj} x[m+di, n+dj] * Keff[2 py + px][.., di+1, dj+1] with
    Keff[py,px][a', b'] = sum_{a,b} W[a,b] * g[a - (py - 2 di) + 1, b - (px - 2 dj) + 1], g = flip(4 f), di = a' - 1.
    Returns [4, O, I, 3, 3] (float64 accumulation, fp32 result)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3 and tuple(resample_filter.shape) == (4, 4)
    w = weight.detach().to(torch.float64)
    g = (resample_filter.detach().to(torch.float64) * 4.0).flip([0, 1]).to(w.device)
    out = torch.zeros([4, o, i, 3, 3], dtype=torch.float64, device=w.device)
    for py in range(2):
        for px in range(2):
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    ty, tx = py - 2 * di, px - 2 * dj
                    for a in range(3):
                        for b in range(3):
                            u, v = a - ty + 1, b - tx + 1
                            if 0 <= u < 4 and 0 <= v < 4:
                                out[2 * py + px, :, :, di + 1, dj + 1] += w[:, :, a, b] * g[u, v]
    return out.to(torch.float32)


def pack_conv_weight_h3_up2_phases(weight: torch.Tensor, resample_filter: torch.Tensor) -> torch.Tensor:
    """The four phase kernels of :func:`fold_up2_fir`, each in the pack_conv_weight_h3 format, back to back
    (operand of nb_modconv3x3_up2_small_h3)."""
    k = fold_up2_fir(weight, resample_filter)
    return torch.stack([pack_conv_weight_h3(k[ph]) for ph in range(4)]).contiguous()


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
    with _on(x.device):
        _lib.check(_lib.lib().nb_pack_h2f8_f32(_p(x.contiguous()), c1, _p(None if x2 is None else x2.contiguous()), c2,
                                               _p(sc), _p(out), n, h * w, _stream(x)), "pack_h2f8")
    return out


F6_CH = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]      # channel of a 16-channel chunk behind field pair i ("f6" format)
_E2M3_GRID = [0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
              2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5]


def e2m3_encode(t: torch.Tensor) -> torch.Tensor:
    """float -> 6-bit e2m3 codes (sign | 2 exponent bits | 3 mantissa bits), round to nearest even, saturating at +-7.5
    (what v_cvt_scalef32_2xpk16_fp6_f32 does: tools/microbench/mfma_f6_check.hip)."""
    a = t.abs().clamp(max=7.5)
    step = torch.where(a < 2, 0.125, torch.where(a < 4, 0.25, 0.5))
    q = torch.round(a / step) * step
    code = torch.where(q < 2, q * 8, torch.where(q < 4, 8 + q * 4, 16 + q * 2)).to(torch.int64)
    return code | (torch.signbit(t).to(torch.int64) << 5)


def e2m3_decode(code: torch.Tensor) -> torch.Tensor:
    grid = torch.tensor(_E2M3_GRID, dtype=torch.float32, device=code.device)
    v = grid[(code & 31).long()]
    return torch.where((code & 32) != 0, -v, v)


def f6_block_exponent(m: torch.Tensor) -> torch.Tensor:
    """Exponent e of a block's scale 2^e: the exponent of the block's largest magnitude minus 2 (0 for an all-zero block)."""
    e = torch.floor(torch.log2(m.clamp(min=2.0 ** -100)))
    return torch.where(m > 0, e - 2, torch.zeros_like(e))


def pack_f6_fields(fields: torch.Tensor) -> torch.Tensor:
    """[..., 32] six-bit codes -> [..., 24] bytes (little-endian bit stream, field f at bits 6f .. 6f+5)."""
    f = fields.to(torch.int64).reshape(*fields.shape[:-1], 8, 4)
    w = f[..., 0] | (f[..., 1] << 6) | (f[..., 2] << 12) | (f[..., 3] << 18)
    return torch.stack([w & 255, (w >> 8) & 255, (w >> 16) & 255], dim=-1).reshape(*fields.shape[:-1], 24).to(torch.uint8)


def unpack_f6_fields(b: torch.Tensor) -> torch.Tensor:
    """[..., 24] bytes -> [..., 32] six-bit codes."""
    t = b.to(torch.int64).reshape(*b.shape[:-1], 8, 3)
    w = t[..., 0] | (t[..., 1] << 8) | (t[..., 2] << 16)
    return torch.stack([w & 63, (w >> 6) & 63, (w >> 12) & 63, (w >> 18) & 63], dim=-1).reshape(*b.shape[:-1], 32)


def pack_conv_weight_h3f6(weight: torch.Tensor) -> torch.Tensor:
    """[O,I,3,3] fp32 -> the "f6" weight format: container of pack_conv_weight_h3; the two lo slots of a (chunk, tap, c_out) hold 32
    e2m3 fields (field 2i = w[ch(i)] / Sw, field 2i+1 = (w - f16(w))[ch(i)] * 2^11 / Sw), the byte of Sw * 2^-11 and zeros
    (include/neube_hip.h)."""
    o, i, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().to(torch.float32)
    nch, op = (i + 15) // 16, (o + 63) // 64 * 64
    wp = torch.zeros([nch * 16, 3, 3, op], dtype=torch.float32, device=w.device)
    wp[:i, :, :, :o] = w.permute(1, 2, 3, 0)
    hi = wp.to(torch.float16)
    lo2048 = (wp - hi.to(torch.float32)) * 2048.0
    out = torch.zeros([nch, 3, 3, 2, 2, op, 16], dtype=torch.uint8, device=w.device)
    hi_b = hi.reshape(nch, 2, 8, 3, 3, op).permute(0, 3, 4, 1, 5, 2).contiguous().view(torch.uint8)      # [chunk,ky,kx,cg,o,16]
    out[:, :, :, :, 0] = hi_b.reshape(nch, 3, 3, 2, op, 16)
    a = wp.reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)                  # [chunk,ky,kx,o,16 ch]
    b = lo2048.reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)
    e = f6_block_exponent(torch.maximum(a.abs(), b.abs()).amax(dim=-1, keepdim=True))
    sw = torch.exp2(e)
    ch = torch.tensor(F6_CH, device=w.device)
    fields = torch.stack([e2m3_encode(a[..., ch] / sw), e2m3_encode(b[..., ch] / sw)], dim=-1).reshape(nch, 3, 3, op, 32)
    by = pack_f6_fields(fields)                                               # [chunk,ky,kx,o,24]
    out[:, :, :, 0, 1] = by[..., :16]
    out[:, :, :, 1, 1, :, :8] = by[..., 16:]
    out[:, :, :, 1, 1, :, 8] = (e[..., 0] + (127 - 11)).clamp(0, 254).to(torch.uint8)
    return out.view(torch.float16).reshape(nch, 3, 3, 2, 2, op, 8)


def pack_h2f6(x, scale=None, x2=None):
    """fp32 NCHW (x ++ x2) * scale[n,c] -> the "f6" activation format (same container as H2)."""
    _dev(x, "x")
    n, c1, h, w = x.shape
    c2 = 0 if x2 is None else x2.shape[1]
    out = torch.empty(h2_shape(n, c1 + c2, h, w), dtype=torch.float16, device=x.device)
    sc = None if scale is None else scale.contiguous()
    with _on(x.device):
        _lib.check(_lib.lib().nb_pack_h2f6_f32(_p(x.contiguous()), c1, _p(None if x2 is None else x2.contiguous()), c2,
                                               _p(sc), _p(out), n, h * w, _stream(x)), "pack_h2f6")
    return out


def unpack_h2f6(t: torch.Tensor, c: int):
    """f6-format tensor [n, c8, 2, h, w, 8] -> (hi f32 [n,c,h,w], xl*2^11 decoded [n,c,h,w], x decoded [n,c,h,w], scale [n,c/16,h,w]):
    the inverse of the format, for tests."""
    nn, c8, _, h, w_, _ = t.shape
    hi = t[:, :, 0].float().permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
    lo = t[:, :, 1].contiguous().view(torch.uint8).reshape(nn, c8 // 2, 2, h, w_, 16)
    by = torch.cat([lo[:, :, 0], lo[:, :, 1, ..., :8]], dim=-1)              # [n, chunk, h, w, 24]
    codes = unpack_f6_fields(by).reshape(nn, c8 // 2, h, w_, 16, 2)
    sc = torch.exp2(lo[:, :, 1, ..., 8].float() - 127.0)                     # [n, chunk, h, w]
    dec = e2m3_decode(codes) * sc[..., None, None]
    inv = torch.argsort(torch.tensor(F6_CH, device=t.device))
    dec = dec[..., inv, :]                                                   # field-pair order -> channel order
    xl = dec[..., 0].permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
    xv = dec[..., 1].permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
    return hi, xl, xv, sc


def h2_shape(n, c, h, w):
    return [n, (c + 7) // 8, 2, h, w, 8]


def pack_h2(x, scale=None, x2=None):
    """fp32 NCHW (optionally the channel concatenation of x and x2) * scale[n,c] -> H2 f16 tensor."""
    _dev(x, "x")
    n, c1, h, w = x.shape
    c2 = 0 if x2 is None else x2.shape[1]
    out = torch.empty(h2_shape(n, c1 + c2, h, w), dtype=torch.float16, device=x.device)
    sc = None if scale is None else scale.contiguous()
    with _on(x.device):
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
    with _on(x_h2.device):
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
    with _on(x_h2.device):
        _lib.check(_lib.lib().nb_modconv3x3_up2_h3(_p(x_h2), c_in, _p(w_h3), _p(dcoefs.contiguous()), _p(noise), ns,
                                                   _p(bias.contiguous()), _p(y), n, h, w, c_out, alpha, float(act_gain),
                                                   float(-1 if act_clamp is None else act_clamp), _stream(x_h2)),
                   "modconv3x3_up2_h3")
    return y
