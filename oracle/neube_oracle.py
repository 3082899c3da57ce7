"""CPU oracle for the NeuBE generator forward path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path
(``brushstroke_engine_amd``) never routes through it and fails loudly when the HIP library is
missing.

What it is: a restatement, in plain ``torch`` CPU tensor ops (fp32 by default, fp64 on request), of
the reference algorithm for the hot path named in BASELINE.json.  Every function cites the
reference file:line it follows (paths relative to /root/reference;
``SG/`` = ``thirdparty/stylegan2_ada_pytorch/``).  Third-party arithmetic under the reference is
PyTorch itself (``F.conv2d``, ``F.conv_transpose2d``, ``addmm``, ``softmax``; the reference pins
torch 1.7/1.8, README.md:25-31) -- the same primitives are used here, so the oracle and the
reference differ only by float re-association.

Pinning: the reference has no tests / golden vectors of its own.  The oracle is pinned
against outputs of the reference itself, generated in the build container by importing
``/root/reference`` (``tests/golden/make_golden.py``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def _t(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a)).to(dtype)


# ----------------------------------------------------------------------------------------------
# Operators (L1)
# ----------------------------------------------------------------------------------------------

def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """``SG/torch_utils/ops/bias_act.py:93-123`` (_bias_act_ref): +bias -> act -> *gain -> clamp.

    Defaults per ``activation_funcs`` (bias_act.py:22-32).  Native twin: ``bias_act.cu:23-147`` (same order,
    fp32 internal).  Plain torch ops, so torch.autograd supplies the first- and second-order gradients the
    reference's ``BiasActCuda`` / ``BiasActCudaGrad`` (bias_act.py:145-204) compute in closed form.
    (One deliberate difference from the reference's *CUDA* gradient: for act='linear' with a clamp the CUDA
    path passes no ``yref`` and so never masks the clamped region, bias_act.py:165 + bias_act.cu:137-138; the
    ``_ref`` path - followed here and by the golden vectors - does mask it.)
    """
    spec = {"linear": (0.0, 1.0), "relu": (0.0, SQRT2), "lrelu": (0.2, SQRT2), "tanh": (0.0, 1.0),
            "sigmoid": (0.0, 1.0), "elu": (0.0, 1.0), "selu": (0.0, 1.0), "softplus": (0.0, 1.0),
            "swish": (0.0, SQRT2)}[act]
    alpha = float(spec[0] if alpha is None else alpha)
    gain = float(spec[1] if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if b is not None:
        assert b.ndim == 1 and b.shape[0] == x.shape[dim]
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    if act == "lrelu":
        x = F.leaky_relu(x, alpha)
    elif act == "relu":
        x = F.relu(x)
    elif act == "tanh":
        x = torch.tanh(x)
    elif act == "sigmoid":
        x = torch.sigmoid(x)
    elif act == "elu":
        x = F.elu(x)
    elif act == "selu":
        x = F.selu(x)
    elif act == "softplus":
        x = F.softplus(x)
    elif act == "swish":
        x = torch.sigmoid(x) * x
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def setup_filter(taps: Sequence[float] = (1, 3, 3, 1), dtype=torch.float32):
    """``SG/torch_utils/ops/upfirdn2d.py:72-116``: <8 taps -> 2-D outer product, normalised to sum 1."""
    f = torch.as_tensor(taps, dtype=torch.float32)
    f = torch.outer(f, f)
    f = f / f.sum()
    return f.to(dtype)


def upfirdn2d(x, f, up=1, down=1, padding=(0, 0, 0, 0), flip_filter=False, gain=1.0):
    """``SG/torch_utils/ops/upfirdn2d.py:168-208`` (_upfirdn2d_ref): zero-stuff, pad/crop, FIR, decimate.

    ``padding`` = [x0, x1, y0, y1]; ``up`` / ``down`` an int or (x, y); ``f`` 2-D, or 1-D = separable taps
    (row pass then column pass, :200-203).  Native twin: ``upfirdn2d.cu:97-200`` (small kernel) / ``:29-92``
    (large kernel).  Plain torch ops: autograd gives the gradient the reference obtains by a second upfirdn2d
    (upfirdn2d.py:245-264).
    """
    n, c, h, w = x.shape
    upx, upy = (up, up) if isinstance(up, int) else up
    downx, downy = (down, down) if isinstance(down, int) else down
    px0, px1, py0, py1 = padding
    x = x.reshape(n, c, h, 1, w, 1)
    x = F.pad(x, [0, upx - 1, 0, 0, 0, upy - 1])
    x = x.reshape(n, c, h * upy, w * upx)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    f = f * (gain ** (f.ndim / 2))
    f = f.to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    if f.ndim == 2:
        x = F.conv2d(x, f[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        x = F.conv2d(x, f[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        x = F.conv2d(x, f[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::downy, ::downx]


def conv2d_resample(x, w, f=None, up=1, padding=0, groups=1, flip_weight=True):
    """``SG/torch_utils/ops/conv2d_resample.py:58-154``, the two branches the generator reaches.

    up=1 (:145-147): plain ``F.conv2d`` with symmetric zero padding (cross-correlation when
    ``flip_weight``).  up=2 (:124-142): stride-2 ``conv_transpose2d`` of the (un-flipped) weight with
    padding 0 -> (2H+1)^2, then ``upfirdn2d(f, padding=[1,1,1,1], gain=4)`` -> (2H)^2 (SURVEY note A).
    """
    oc, icg, kh, kw = w.shape
    if up == 1:
        if not flip_weight:
            w = w.flip([2, 3])
        return F.conv2d(x, w, padding=padding, groups=groups)
    assert up == 2 and kh == 3 and kw == 3 and f is not None and f.shape == (4, 4) and padding == 1
    # padding bookkeeping of conv2d_resample.py:103-107,131-136 for k=3, fw=4, up=2, padding=1:
    px0 = padding + (4 + up - 1) // 2 - (kw - 1)     # = 1
    px1 = padding + (4 - up) // 2 - (kw - up)        # = 1
    assert px0 == 1 and px1 == 1
    if groups == 1:
        wt = w.transpose(0, 1)
    else:
        wt = w.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2).reshape(groups * icg, oc // groups, kh, kw)
    # _conv2d_wrapper(transpose=True, flip_weight=(not flip_weight)): conv_transpose2d is a true convolution,
    # so flip_weight=False at the call site (networks.py:384) means: no extra flip here.
    if flip_weight:
        wt = wt.flip([2, 3])
    y1 = F.conv_transpose2d(x, wt, stride=2, padding=0, groups=groups)
    return upfirdn2d(y1, f, padding=(px0, px1, px0, px1), gain=up ** 2)


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """``SG/training/networks.py:30-88`` (fp32 branches).

    fused (:55-64, 78-88): w[n,o,i,ky,kx] = W*s; d = rsqrt(sum(w^2)+1e-8); w *= d; grouped conv with
    groups=N; ``+ noise``.  non-fused (:67-76): x*s -> shared conv -> *d + noise.
    """
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    dcoefs = None
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if not fused_modconv:
        x = x * styles.reshape(n, -1, 1, 1)
        x = conv2d_resample(x, weight, f=resample_filter, up=up, padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = torch.addcmul(noise, x, dcoefs.reshape(n, -1, 1, 1))      # fma.py:15-26
        elif demodulate:
            x = x * dcoefs.reshape(n, -1, 1, 1)
        elif noise is not None:
            x = x + noise
        return x
    if demodulate:
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, ic, kh, kw)
    x = conv2d_resample(x, w, f=resample_filter, up=up, padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


def fully_connected(x, weight, bias, activation="linear", lr_multiplier=1.0):
    """``SG/training/networks.py:109-122``: weight_gain = lr_mul/sqrt(in), bias_gain = lr_mul."""
    w = weight * (lr_multiplier / math.sqrt(weight.shape[1]))
    b = bias
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier
    if activation == "linear" and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    x = x.matmul(w.t())
    return bias_act(x, b, act=activation)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """``SG/training/networks.py:24-26``."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def shifted_const_noise(noise_const, noise_grid, norm_positions):
    """Position-shifted constant noise, explicit form of ``SG/training/networks.py:373-381`` (SURVEY note C).

    The reference calls ``grid_sample(noise[None,None].expand(N), ((grid + pos) % 1)*2-1, bilinear,
    padding_mode='reflection', align_corners=True)``.  With align_corners the sample coordinate is
    ``((g+1)/2)*(r-1)`` and g in [-1,1) never leaves the image, so reflection is the identity and the
    only out-of-range corner is ``floor+1 == r`` with weight 0.  grid[...,0] (= lin[i] + pos[:,0]) is the
    *column* coordinate and grid[...,1] (= lin[j] + pos[:,1]) the *row*, hence the transposition.
    ``tests/test_oracle_golden.py`` checks this against ``F.grid_sample`` itself.
    """
    r = noise_const.shape[0]
    dt = noise_const.dtype
    g = (noise_grid.to(dt) + norm_positions.to(dt).unsqueeze(1).unsqueeze(1)) % 1      # [N,r,r,2]
    g = g * 2 - 1
    cx = ((g[..., 0] + 1) / 2) * (r - 1)      # column coordinate
    cy = ((g[..., 1] + 1) / 2) * (r - 1)      # row coordinate
    x0 = torch.floor(cx)
    y0 = torch.floor(cy)
    wx1 = cx - x0
    wy1 = cy - y0
    wx0 = (x0 + 1) - cx
    wy0 = (y0 + 1) - cy
    x0i = x0.long()
    y0i = y0.long()
    x1i = x0i + 1
    y1i = y0i + 1

    def tap(yi, xi):
        ok = (yi >= 0) & (yi < r) & (xi >= 0) & (xi < r)
        v = noise_const[yi.clamp(0, r - 1), xi.clamp(0, r - 1)]
        return torch.where(ok, v, torch.zeros((), dtype=dt))

    out = tap(y0i, x0i) * (wx0 * wy0) + tap(y0i, x1i) * (wx1 * wy0) \
        + tap(y1i, x0i) * (wx0 * wy1) + tap(y1i, x1i) * (wx1 * wy1)
    return out.unsqueeze(1)                                                        # [N,1,r,r]


def blend(features, alpha, x):
    """``forger/train/stitching.py:24-25`` (BlendedFeatures.blend): alpha*F + (1-alpha)*x."""
    return alpha * features + (1 - alpha) * x


# ----------------------------------------------------------------------------------------------
# Network (L2)
# ----------------------------------------------------------------------------------------------

class OracleGenerator:
    """Restatement of ``SG/training/networks_modified.py`` Generator/SynthesisNetwork (+ layers from
    ``SG/training/networks.py``) for ``architecture='orig'``, ``color_format='triad'``,
    ``color_w_channels=0``, ``noise_mode='const'``, ``force_fp32=True``, ``positional_encoding=None``.
    """

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray], dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: _t(v, dtype) for k, v in state_dict.items()}
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, cfg.c_dim, cfg.w_dim
        self.img_resolution, self.img_channels = cfg.img_resolution, cfg.img_channels
        self.num_ws = cfg.num_ws
        self.filter = setup_filter(cfg.resample_filter, dtype)

    # -- MappingNetwork.forward, networks.py:255-290 (c_dim == 0; eval mode: no w_avg update) --
    def mapping(self, z, c=None, truncation_psi=1, truncation_cutoff=None):
        x = normalize_2nd_moment(_t(z, self.dtype))
        for i in range(self.cfg.mapping_layers):
            x = fully_connected(x, self.sd[f"mapping.fc{i}.weight"], self.sd[f"mapping.fc{i}.bias"],
                                activation="lrelu", lr_multiplier=self.cfg.mapping_lr_multiplier)
        x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            w_avg = self.sd["mapping.w_avg"]
            if truncation_cutoff is None:
                x = w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x

    # -- SynthesisLayer.forward, networks.py:362-391 --
    def layer(self, spec, x, w, norm_noise_positions=None, input_noise=None, fused_modconv=True, taps=None):
        sd, name = self.sd, spec.name
        styles = fully_connected(w, sd[f"{name}.affine.weight"], sd[f"{name}.affine.bias"])
        noise_const = sd[f"{name}.noise_const"] if input_noise is None else _t(input_noise, self.dtype)
        if norm_noise_positions is not None:
            noise_const = shifted_const_noise(noise_const, sd[f"{name}.noise_grid"], norm_noise_positions)
        noise = noise_const * sd[f"{name}.noise_strength"]
        y = modulated_conv2d(x, sd[f"{name}.weight"], styles, noise=noise, up=spec.up, padding=1,
                             resample_filter=self.filter, flip_weight=(spec.up == 1), fused_modconv=fused_modconv)
        clamp = self.cfg.conv_clamp
        y = bias_act(y, sd[f"{name}.bias"], act="lrelu", gain=SQRT2, clamp=clamp)
        if taps is not None:
            taps[f"{name}.styles"] = styles
            taps[f"{name}.out"] = y
        return y

    # -- ToRGBColorTriadLayer.forward, networks.py:451-485 (color_w_channels == 0, 'triad') --
    def torgb(self, x, w, taps=None):
        sd, t = self.sd, self.cfg.torgb_name
        c = x.shape[1]
        scaled = fully_connected(w, sd[f"{t}.affine.weight"], sd[f"{t}.affine.bias"])
        colors = bias_act(scaled[:, 0:9], sd[f"{t}.color_bias"], dim=1, act="tanh").reshape(-1, 3, 3)
        styles = scaled[:, 9:] * (1 / math.sqrt(c))
        y = modulated_conv2d(x, sd[f"{t}.weight"], styles, demodulate=False)
        y = bias_act(y, sd[f"{t}.bias"], clamp=self.cfg.conv_clamp)
        uvs = torch.softmax(y[:, :3], dim=1)
        img = torch.sum(uvs.unsqueeze(1) * colors.unsqueeze(-1).unsqueeze(-1), dim=2)
        if taps is not None:
            taps["torgb.logits"] = y
        return img, {"colors": colors, "uvs": uvs}

    # -- SynthesisNetwork.forward, networks_modified.py:123-223 (+ SynthesisBlock.forward networks.py:630-680) --
    def synthesis(self, ws, geom_feature, return_debug_data=False, return_features=None, blended_features=None,
                  noise_buffers=None, norm_noise_positions=None, fused_modconv=True, taps=None):
        cfg = self.cfg
        ws = _t(ws, self.dtype)
        assert ws.shape[1:] == (cfg.num_ws, cfg.w_dim)
        return_features = return_features or []
        blended_features = blended_features or {}
        if norm_noise_positions is not None:
            norm_noise_positions = _t(norm_noise_positions, self.dtype)
        n = ws.shape[0]
        debug = {}
        x = img = None
        geo_idx = 0
        layers = {l.name: l for l in cfg.layers}
        for res in cfg.block_resolutions:
            bname = f"synthesis.b{res}"
            nb = noise_buffers or {}
            if res == 4:
                x = self.sd["synthesis.b4.const"].unsqueeze(0).repeat([n, 1, 1, 1])
            else:
                l0 = layers[f"{bname}.conv0"]
                x = self.layer(l0, x, ws[:, l0.w_index], norm_noise_positions, nb.get(f"b{res}.conv0.noise_const"),
                               fused_modconv, taps)
            l1 = layers[f"{bname}.conv1"]
            x = self.layer(l1, x, ws[:, l1.w_index], norm_noise_positions, nb.get(f"b{res}.conv1.noise_const"),
                           fused_modconv, taps)
            is_last = res == cfg.img_resolution
            if is_last:
                img, triad = self.torgb(x, ws[:, cfg.torgb_w_index], taps)
                if return_debug_data:
                    debug.update(triad)
            if res in return_features:
                debug[f"features{res}_preblend"] = x
            if res in blended_features:
                bf = blended_features[res]
                x = blend(_t(bf["features"], self.dtype), _t(bf["alpha"], self.dtype), x)
                if is_last:
                    img, triad = self.torgb(x, ws[:, cfg.torgb_w_index], taps)
                    debug.update(triad)
            if res in return_features:
                debug[f"features{res}"] = x
            if res in cfg.geom_feature_resolutions:
                x = torch.cat([x, _t(geom_feature[geo_idx], self.dtype)], dim=1)
                geo_idx += 1
        if len(debug) > 0:
            return img, debug
        return img

    # -- Generator.forward_pre_mapped, networks_modified.py:346-365 --
    def forward_pre_mapped(self, ws, geom_feature, positions=None, return_debug_data=False, return_features=None,
                           blended_features=None, noise_buffers=None, **kw):
        norm_positions = None
        if positions is not None:
            positions = torch.as_tensor(np.asarray(positions)).to(torch.int64)
            if self.dtype == torch.float64:
                norm_positions = (positions % self.img_resolution).to(torch.float64) / (self.img_resolution - 1)
            else:
                norm_positions = (positions % self.img_resolution) / (self.img_resolution - 1)   # float32 true division
        res = self.synthesis(ws, geom_feature, return_debug_data=return_debug_data, return_features=return_features,
                             blended_features=blended_features, noise_buffers=noise_buffers,
                             norm_noise_positions=norm_positions, **kw)
        if return_debug_data or return_features:
            img, debug = res
            if return_debug_data:
                debug["ws"] = _t(ws, self.dtype)
            return img, debug
        return res

    # -- Generator.forward, networks_modified.py:367-400 (style_mixing_prob == 0) --
    def forward(self, z, c, geom_feature, positions=None, noise_buffers=None, truncation_psi=1,
                truncation_cutoff=None, return_debug_data=False, return_features=None, blended_features=None, **kw):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.forward_pre_mapped(ws, geom_feature, positions=positions, return_debug_data=return_debug_data,
                                       return_features=return_features, blended_features=blended_features,
                                       noise_buffers=noise_buffers, **kw)

    __call__ = forward


# ----------------------------------------------------------------------------------------------
# Engine compositing (L3, row a15)
# ----------------------------------------------------------------------------------------------

def map_style_s(sfactor, uvs):
    """``StyleUVSMapper._map_style_s`` (forger/ui/mapper.py:52-72): S' = min(sfactor*S, 1); (U', V') = (U, V) scaled
    so that U'+V'+S' = 1 (zero where 1-S' <= 1e-6)."""
    U, V, S = uvs[:, 0:1], uvs[:, 1:2], uvs[:, 2:3]
    Sp = torch.clamp(sfactor * S, max=1.0)
    delta = 1 - Sp
    f = torch.where(delta <= 0.000001, torch.zeros_like(delta), delta / (U + V))
    return torch.cat([f * U, f * V, Sp], dim=1)


def triad_composite(uvs, colors, render_mode="clear", user_colors=None, sfactor=None):
    """``forger/ui/brush.py:763-792`` (TriadGanPaintEngine._render_stroke_torch, enable_uvs_mapping=False).

    default_colors = (colors+1)/2 [N,3(rgb),3(k)]; ``user_colors`` (same layout, NaN = keep default)
    replace them (GanBrushOptions.prepare_colors); stroke = sum_k uvs_k*color_k; 'clear': alpha = u+v,
    'full': alpha = 1.  Returns RGBA float [N,4,R,R] in [0,1].
    """
    default_colors = (colors + 1) / 2.0
    if sfactor is not None:                        # brush.py:773-774, enable_uvs_mapping
        uvs = map_style_s(sfactor, uvs)
    if user_colors is not None:
        uc = _t(user_colors, colors.dtype)
        default_colors = torch.where(torch.isnan(uc), default_colors, uc)
    stroke = torch.sum(uvs.unsqueeze(1) * default_colors.unsqueeze(-1).unsqueeze(-1), dim=2)
    if render_mode == "clear":
        alpha = torch.sum(uvs[:, 0:2], dim=1, keepdim=True)
    elif render_mode == "full":
        alpha = torch.ones_like(stroke[:, :1])
    else:
        raise RuntimeError(f"Unknown render mode for TriadGanPaintEngine: {render_mode}")
    return torch.cat([stroke, alpha], dim=1)


def rgba_to_uint8(rgba):
    """``forger/ui/brush.py:377`` path: (x*255).clip(0,255) -> uint8 truncation."""
    return (rgba * 255).clamp(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------------
# Independent direct-loop conv (no torch conv) for cross-checking the primitives on tiny cases
# ----------------------------------------------------------------------------------------------

def direct_modconv_numpy(x, weight, styles, up=1, demodulate=True, f=None):
    """Textbook evaluation of SURVEY note A in float64 numpy loops (small shapes only).

    up=1: y[n,o,y,x] = sum_{c,ky,kx} xpad[n,c,y+ky,x+kx] * w[n,o,c,ky,kx]
    up=2: y1[n,o,2i+a,2j+b] += x[n,c,i,j]*w[n,o,c,a,b];  y = corr(pad(y1,1), 4*f)
    """
    x = np.asarray(x, np.float64)
    weight = np.asarray(weight, np.float64)
    styles = np.asarray(styles, np.float64)
    n, c, h, wd = x.shape
    oc = weight.shape[0]
    w = weight[None] * styles[:, None, :, None, None]
    if demodulate:
        w = w / np.sqrt((w ** 2).sum(axis=(2, 3, 4), keepdims=True) + 1e-8)
    if up == 1:
        xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
        y = np.zeros((n, oc, h, wd))
        for ky in range(3):
            for kx in range(3):
                y += np.einsum("nchw,noc->nohw", xp[:, :, ky:ky + h, kx:kx + wd], w[:, :, :, ky, kx])
        return y
    y1 = np.zeros((n, oc, 2 * h + 1, 2 * wd + 1))
    for a in range(3):
        for b in range(3):
            y1[:, :, a:a + 2 * h:2, b:b + 2 * wd:2] += np.einsum("nchw,noc->nohw", x, w[:, :, :, a, b])
    f = np.asarray(f, np.float64) * 4.0
    yp = np.pad(y1, ((0, 0), (0, 0), (1, 1), (1, 1)))
    y = np.zeros((n, oc, 2 * h, 2 * wd))
    for p in range(4):
        for q in range(4):
            y += f[3 - p, 3 - q] * yp[:, :, p:p + 2 * h, q:q + 2 * wd]
    return y
