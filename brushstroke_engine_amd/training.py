"""Differentiable generator: the same network as :mod:`networks`, evaluated layer by layer through the
operator layer's autograd functions instead of the fused inference launches, so that gradients reach every parameter,
the latents and the noise inputs - what the reference's training loop (``training_loop_modified.py:50-668``) and its
projection script differentiate through.

Where the arithmetic lives is the same as in the reference:
  * ``modulated_conv2d`` / ``bias_act`` / ``upfirdn2d`` (the reference's native plugins + cuDNN via ``conv2d_gradfix``):
    hand-written HIP kernels, forward and backward (:mod:`ops`, ``csrc/nb_grad.hip``);
  * fully connected layers (``torch.addmm``, networks.py:118-120), the second-moment normalisation, ``softmax``, the triad
    composite and the position shift of the constant noise (``grid_sample``, networks.py:373-381): torch ops, as in the
    reference itself.
The 1x1 modulated conv of ``ToRGBColorTriadLayer`` (networks.py:466-470) runs as the centre tap of a 3x3 one.

This is the first, unfused training-mode path: correct gradients, not a tuned step.
"""
from __future__ import annotations

import os

import collections.abc

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .config import GeneratorConfig
from .weights import StateDict, validate_state_dict

_TRAINABLE_SUFFIXES = (".weight", ".bias", ".noise_strength", ".const", ".color_bias")


class TrainableGenerator(torch.nn.Module):
    """Parameters under the reference's ``state_dict`` names (``mapping.fc0.weight``, ``synthesis.b64.conv0.weight`` ...,
    dots replaced by ``__`` inside the module); buffers (noise_const, noise_grid, resample_filter, w_avg) likewise."""

    def __init__(self, cfg: GeneratorConfig, state_dict: StateDict, device="cuda"):
        super().__init__()
        validate_state_dict(cfg, state_dict)
        self.cfg = cfg
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, cfg.c_dim, cfg.w_dim
        self.img_resolution, self.img_channels, self.num_ws = cfg.img_resolution, cfg.img_channels, cfg.num_ws
        self._names: Dict[str, str] = {}
        for k, v in state_dict.items():
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32)))
            attr = k.replace(".", "__")
            self._names[k] = attr
            if k.endswith(_TRAINABLE_SUFFIXES):
                self.register_parameter(attr, torch.nn.Parameter(t))
            else:
                self.register_buffer(attr, t)
        self.to(device)

    def p(self, key: str) -> torch.Tensor:
        return getattr(self, self._names[key])

    def named_reference_parameters(self):
        """(reference key, Parameter) pairs."""
        for k, attr in self._names.items():
            t = getattr(self, attr)
            if isinstance(t, torch.nn.Parameter):
                yield k, t

    # -- FullyConnectedLayer.forward, networks.py:109-122 --
    def _fc(self, x, prefix: str, activation: str = "linear", lr_multiplier: float = 1.0):
        w = self.p(prefix + ".weight")
        b = self.p(prefix + ".bias")
        w = w * (lr_multiplier / math.sqrt(w.shape[1]))
        if lr_multiplier != 1:
            b = b * lr_multiplier
        if activation == "linear":
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return ops.bias_act(x.matmul(w.t()).contiguous(), b.contiguous(), act=activation)

    # -- MappingNetwork.forward, networks.py:255-290 (c_dim == 0, no truncation) --
    w_avg_beta = 0.995                                  # MappingNetwork default, networks.py:227

    def mapping(self, z, c=None, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = z.to(torch.float32)
        x = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
        for i in range(self.cfg.mapping_layers):
            x = self._fc(x, f"mapping.fc{i}", "lrelu", self.cfg.mapping_lr_multiplier)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:      # networks.py:274-276
            with torch.no_grad():
                w_avg = self.p("mapping.w_avg")
                w_avg.copy_(x.detach().mean(dim=0).lerp(w_avg, self.w_avg_beta))
        x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:                                                          # networks.py:283-289
            w_avg = self.p("mapping.w_avg")
            if truncation_cutoff is None:
                x = w_avg.lerp(x, truncation_psi)
            else:
                x = torch.cat([w_avg.lerp(x[:, :truncation_cutoff], truncation_psi), x[:, truncation_cutoff:]], dim=1)
        return x

    # -- SynthesisLayer.forward, networks.py:362-391 --
    def _layer(self, spec, x, w, norm_pos, noise_mode):
        name = spec.name
        styles = self._fc(w, name + ".affine")
        noise = None
        if noise_mode == "random":
            noise = torch.randn([x.shape[0], 1, spec.block_res, spec.block_res], device=x.device) * self.p(name + ".noise_strength")
        elif noise_mode == "const":
            nc = self.p(name + ".noise_const")
            if norm_pos is not None:
                grid = (self.p(name + ".noise_grid") + norm_pos.unsqueeze(1).unsqueeze(1)) % 1 * 2 - 1
                nc = F.grid_sample(nc[None, None].expand(x.shape[0], -1, -1, -1), grid, padding_mode="reflection", align_corners=True)
            noise = nc * self.p(name + ".noise_strength")
            if noise.ndim == 2:
                noise = noise[None, None]
        y = ops.modulated_conv2d(x.contiguous(), self.p(name + ".weight"), styles.contiguous(),
                                 noise=None if noise is None else noise.contiguous(), up=spec.up, padding=1,
                                 resample_filter=self.p(name + ".resample_filter"), flip_weight=(spec.up == 1))
        return ops.bias_act(y, self.p(name + ".bias"), act="lrelu", gain=math.sqrt(2), clamp=self.cfg.conv_clamp)

    # -- ToRGBColorTriadLayer.forward, networks.py:451-485 --
    def _torgb(self, x, w):
        t = self.cfg.torgb_name
        c = x.shape[1]
        scaled = self._fc(w, t + ".affine")
        colors = ops.bias_act(scaled[:, 0:9].contiguous(), self.p(t + ".color_bias"), dim=1, act="tanh").reshape(-1, 3, 3)
        styles = (scaled[:, 9:] * (1 / math.sqrt(c))).contiguous()
        w3 = F.pad(self.p(t + ".weight"), (1, 1, 1, 1))                     # 1x1 kernel as the centre tap of a 3x3 one
        y = ops.modulated_conv2d(x.contiguous(), w3, styles, up=1, padding=1, demodulate=False, flip_weight=True)
        y = ops.bias_act(y, self.p(t + ".bias"), clamp=self.cfg.conv_clamp)
        uvs = torch.softmax(y[:, :3], dim=1)
        img = torch.sum(uvs.unsqueeze(1) * colors.unsqueeze(-1).unsqueeze(-1), dim=2)
        return img, {"colors": colors, "uvs": uvs}

    # -- SynthesisNetwork.forward, networks_modified.py:123-223 ('orig' architecture, triad colours) --
    def synthesis(self, ws, geom_feature, norm_noise_positions=None, noise_mode="random", return_debug_data=False):
        """``noise_mode`` defaults to 'random' like ``SynthesisLayer.forward`` (networks.py:362): training draws fresh
        per-layer noise every step; 'const' is what inference (and the golden-vector tests) pass explicitly."""
        cfg = self.cfg
        ws = ws.to(torch.float32)
        n = ws.shape[0]
        geom_feature = list(geom_feature) if isinstance(geom_feature, (list, tuple)) else [geom_feature]
        layers = {l.name: l for l in cfg.layers}
        x = img = None
        triad = {}
        geo_idx = 0
        for res in cfg.block_resolutions:
            b = f"synthesis.b{res}"
            if res == 4:
                x = self.p(b + ".const").unsqueeze(0).repeat([n, 1, 1, 1])
            else:
                sp = layers[b + ".conv0"]
                x = self._layer(sp, x, ws[:, sp.w_index], norm_noise_positions, noise_mode)
            sp = layers[b + ".conv1"]
            x = self._layer(sp, x, ws[:, sp.w_index], norm_noise_positions, noise_mode)
            if res == cfg.img_resolution:
                img, triad = self._torgb(x, ws[:, cfg.torgb_w_index])
            if res in cfg.geom_feature_resolutions:
                x = torch.cat([x, geom_feature[geo_idx].to(torch.float32)], dim=1)     # networks_modified.py:218-219
                geo_idx += 1
        return (img, triad) if return_debug_data else img

    def forward(self, z, c, geom_feature, positions=None, noise_mode="random", return_debug_data=False,
                truncation_psi=1, truncation_cutoff=None, style_mixing_prob=0):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        if style_mixing_prob > 0:                                          # networks_modified.py:385-394
            from .networks import mix_styles
            ws = mix_styles(self.mapping, ws, z, c, style_mixing_prob, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        norm_pos = None
        if positions is not None:                                          # networks_modified.py:351-353
            norm_pos = (positions % self.img_resolution).to(torch.float32) / (self.img_resolution - 1)
        res = self.synthesis(ws, geom_feature, norm_noise_positions=norm_pos, noise_mode=noise_mode,
                             return_debug_data=return_debug_data)
        if return_debug_data:
            res[1]["ws"] = ws
        return res


class TrainableDiscriminator(torch.nn.Module):
    """``training/networks.py:788-1012`` for ``architecture='resnet'``, ``c_dim=0``, fp32: residual blocks
    (fromrgb 1x1 at the input resolution; conv0 3x3; conv1 3x3 with FIR + stride-2 down-sampling; skip 1x1 down-sampling,
    both halves scaled by sqrt(1/2)), minibatch-stddev feature, 3x3 conv, two fully connected layers -> one logit.
    Every ``Conv2dLayer`` (networks.py:125-173) is ``conv2d_resample`` + ``bias_act`` on the differentiable HIP operators
    (gradients of any order: the R1 penalty differentiates the input gradient once more); parameters carry the
    reference's ``state_dict`` names."""

    def __init__(self, state_dict: StateDict, img_resolution: int, img_channels: int, channel_base: int = 32768,
                 channel_max: int = 512, conv_clamp: Optional[float] = None, mbstd_group_size: Optional[int] = 4,
                 mbstd_num_channels: int = 1, resample_filter=(1, 3, 3, 1), device="cuda"):
        super().__init__()
        self.img_resolution, self.img_channels, self.conv_clamp = img_resolution, img_channels, conv_clamp
        self.mbstd_group_size, self.mbstd_num_channels = mbstd_group_size, mbstd_num_channels
        log2 = int(math.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(log2, 2, -1)]
        self.channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        self._names: Dict[str, str] = {}
        for k, v in state_dict.items():
            if k.endswith("resample_filter"):
                continue
            attr = k.replace(".", "__")
            self._names[k] = attr
            self.register_parameter(attr, torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32)))))
        self.register_buffer("fir", ops.setup_filter(resample_filter))
        self.to(device)

    def p(self, key: str):
        return getattr(self, self._names[key]) if key in self._names else None

    def named_reference_parameters(self):
        for k, attr in self._names.items():
            yield k, getattr(self, attr)

    # -- Conv2dLayer.forward (networks.py:163-173) with conv2d_resample's down-sampling branches (conv2d_resample.py:96-113) --
    def _conv(self, x, prefix: str, k: int, down: int = 1, activation: str = "lrelu", gain: float = 1.0, clamp=None):
        w = self.p(prefix + ".weight")
        w = w * (1 / math.sqrt(w.shape[1] * k * k))
        pad = k // 2
        if down == 1:
            x = ops.conv2d(x, w, stride=1, padding=pad)
        else:
            fw = self.fir.shape[1]
            p0, p1 = pad + (fw - down + 1) // 2, pad + (fw - down) // 2
            if k == 1:
                x = ops.upfirdn2d(x, self.fir, down=down, padding=[p0, p1, p0, p1])
                x = ops.conv2d(x, w, stride=1, padding=0)
            elif k == 3 and down == 2 and (p0, p1) == (2, 2):
                x = ops.conv2d_down2(x, w, self.fir)
            else:
                x = ops.upfirdn2d(x, self.fir, padding=[p0, p1, p0, p1])
                x = ops.conv2d(x, w, stride=down, padding=0)
        act_gain = (math.sqrt(2) if activation == "lrelu" else 1.0) * gain
        return ops.bias_act(x, self.p(prefix + ".bias"), act=activation, gain=act_gain, clamp=None if clamp is None else clamp * gain)

    def _fc(self, x, prefix: str, activation: str = "linear"):
        w = self.p(prefix + ".weight")
        w = w * (1 / math.sqrt(w.shape[1]))
        b = self.p(prefix + ".bias")
        if activation == "linear":
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return ops.bias_act(x.matmul(w.t()).contiguous(), b, act=activation)

    def forward(self, img, c=None, sub_batches: int = 1):
        """``sub_batches`` > 1: ``img`` is that many independent batches stacked along dim 0 (e.g. generated and real images of
        one discriminator phase); the minibatch-stddev statistics -- the one place where samples of a batch meet -- are taken per
        sub-batch, so the logits equal those of separate calls."""
        x = None
        img = img.to(torch.float32).contiguous()
        for res in self.block_resolutions:
            b = f"b{res}"
            if x is None:
                x = self._conv(img, b + ".fromrgb", 1, clamp=self.conv_clamp)
            y = self._conv(x, b + ".skip", 1, down=2, activation="linear", gain=math.sqrt(0.5))
            x = self._conv(x, b + ".conv0", 3, clamp=self.conv_clamp)
            x = self._conv(x, b + ".conv1", 3, down=2, gain=math.sqrt(0.5), clamp=self.conv_clamp)
            x = y + x
        if self.mbstd_num_channels > 0:                                   # MinibatchStdLayer, networks.py:868-885
            nt, ch, h, w = x.shape
            assert nt % sub_batches == 0
            n = nt // sub_batches
            g = n if self.mbstd_group_size is None else min(self.mbstd_group_size, n)
            f_, c_ = self.mbstd_num_channels, ch // self.mbstd_num_channels
            y = x.reshape(sub_batches, g, -1, f_, c_, h, w)
            y = y - y.mean(dim=1, keepdim=True)
            y = (y.square().mean(dim=1) + 1e-8).sqrt().mean(dim=[3, 4, 5])            # [sub, n/g, F]
            y = y.reshape(sub_batches, 1, -1, f_, 1, 1).expand(sub_batches, g, -1, f_, h, w).reshape(nt, f_, h, w)
            x = torch.cat([x, y], dim=1)
        x = self._conv(x.contiguous(), "b4.conv", 3, clamp=self.conv_clamp)
        x = self._fc(x.flatten(1), "b4.fc", "lrelu")
        return self._fc(x, "b4.out")


def random_discriminator_state_dict(img_resolution: int, img_channels: int, channel_base: int = 32768, channel_max: int = 512,
                                    mbstd_num_channels: int = 1, seed: int = 0, bias_std: float = 0.0) -> Dict[str, np.ndarray]:
    """Parameters of a freshly initialised reference discriminator (resnet, c_dim=0): N(0,1) weights, zero biases
    (``Conv2dLayer`` networks.py:150-153, ``FullyConnectedLayer`` :104-105); ``bias_std`` > 0 randomises the biases for tests."""
    rs = np.random.RandomState(seed)
    log2 = int(math.log2(img_resolution))
    ch = {res: min(channel_base // res, channel_max) for res in [2 ** i for i in range(log2, 1, -1)]}
    sd: Dict[str, np.ndarray] = {}

    def conv(name, ci, co, k, bias=True):
        sd[name + ".weight"] = rs.randn(co, ci, k, k).astype(np.float32)
        if bias:
            sd[name + ".bias"] = (bias_std * rs.randn(co)).astype(np.float32)

    for res in [2 ** i for i in range(log2, 2, -1)]:
        tmp, out = ch[res], ch[res // 2]
        if res == img_resolution:
            conv(f"b{res}.fromrgb", img_channels, tmp, 1)
        conv(f"b{res}.conv0", tmp, tmp, 3)
        conv(f"b{res}.conv1", tmp, out, 3)
        conv(f"b{res}.skip", tmp, out, 1, bias=False)
    conv("b4.conv", ch[4] + mbstd_num_channels, ch[4], 3)
    sd["b4.fc.weight"] = rs.randn(ch[4], ch[4] * 16).astype(np.float32)
    sd["b4.fc.bias"] = (bias_std * rs.randn(ch[4])).astype(np.float32)
    sd["b4.out.weight"] = rs.randn(1, ch[4]).astype(np.float32)
    sd["b4.out.bias"] = (bias_std * rs.randn(1)).astype(np.float32)
    return sd


class LazyStats(collections.abc.Mapping):
    """Loss statistics of a phase.  Values are kept as detached 0-dim DEVICE tensors and become python floats only when they are
    read -- the reference's ``training_stats.report`` likewise accumulates on the device (torch_utils/training_stats.py:86-101);
    reading a loss value between the forward and the backward pass would stall the host until the device has caught up and
    leave the device idle while the backward pass is being issued.

    A read-only ``Mapping`` over a private dict (NOT a ``dict`` subclass: CPython's fast paths -- ``dict(stats)``, ``{**stats}``,
    ``plain.update(stats)``, ``json.dumps`` -- bypass overridden accessors of a dict subclass and would hand out the raw device
    tensors).  Every access path of the Mapping protocol goes through ``__getitem__`` and yields floats; ``resolve()`` converts
    all values with ONE device-to-host transfer and is what a stats writer should call."""

    def __init__(self, init=()):
        self._d = {}
        self.update(init)

    @staticmethod
    def _f(v):
        return float(v) if torch.is_tensor(v) else v

    def __getitem__(self, k):
        return self._f(self._d[k])

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __setitem__(self, k, v):
        self._d[k] = v

    def update(self, other=(), **kw):
        if isinstance(other, LazyStats):
            self._d.update(other._d)
        else:
            self._d.update(dict(other))
        self._d.update(kw)

    def resolve(self) -> Dict[str, float]:
        """All values as python floats: the device tensors are stacked and fetched in one transfer."""
        keys = [k for k, v in self._d.items() if torch.is_tensor(v)]
        out = {k: v for k, v in self._d.items() if not torch.is_tensor(v)}
        if keys:
            vals = torch.stack([self._d[k].reshape(()).to(torch.float32) for k in keys]).tolist()
            out.update(zip(keys, vals))
        return {k: out[k] for k in self._d}

    def __repr__(self):
        return f"LazyStats({self.resolve()!r})"


class GanLoss:
    """The adversarial phases of ``ForgerLoss.accumulate_gradients`` (loss_modified.py:140-272): non-saturating logistic
    losses for G and D, the R1 penalty on real images and the path-length regulariser of G (second-order gradient of
    the modulated convolution).  ``phase`` in {'Gmain', 'Greg', 'Dmain', 'Dreg', 'Dall'}; gradients are accumulated into
    the parameters' ``.grad`` like the reference does.  The generator runs with fresh random per-layer noise and
    ``style_mixing_prob`` like ``ForgerLoss.run_G`` (loss_modified.py:72-100); ``noise_mode='const'`` is for tests that
    need a deterministic forward.  Each phase switches ``requires_grad`` off on the network it does not optimise, the
    way the reference loop brackets a phase (training_loop_modified.py: ``phase.module.requires_grad_(True)`` ...
    ``requires_grad_(False)``), so a G phase runs none of D's weight-gradient kernels (and DDP reduces none).
    ``augment_pipe`` = the ADA pipeline (:mod:`augment`).  The forger phases: ``Ggeom`` / ``Ggeom-warm`` evaluate
    ``geom_phase_losses`` / ``geom_warmstart_losses`` (strings of :mod:`forger_losses`, e.g. the shipped
    ``'1.0*iou_inv(uvs)'``) on the generator's debug dict against the stroke geometry (loss_modified.py:181-203),
    ``Gmain`` adds ``main_phase_losses``, and :meth:`accumulate_gradients_stitch` is the ``Gstitch`` phase
    (loss_modified.py:108-138: two overlapping crops, composites judged by D, ``stitch_phase_losses``)."""

    def __init__(self, G: TrainableGenerator, D: TrainableDiscriminator, r1_gamma: float = 10.0, pl_batch_shrink: int = 2,
                 pl_decay: float = 0.01, pl_weight: float = 2.0, augment_pipe=None, style_mixing_prob: float = 0.9,
                 noise_mode: str = "random", geom_phase_losses: str = "", main_phase_losses: str = "",
                 geom_warmstart_losses: Optional[str] = None, stitch_phase_losses: str = "", stitcher=None,
                 partial_loss_with_triband_input: bool = False, merge_d_passes: bool = True):
        from .forger_losses import ForgerLosses, RandomStitcher
        self._G, self.r1_gamma = G, r1_gamma
        self.merge_d_passes = merge_d_passes                # 'Dmain': generated + real images as one stacked discriminator batch
        self._D, self.augment_pipe = D, augment_pipe
        self.style_mixing_prob, self.noise_mode = style_mixing_prob, noise_mode
        self.geom_phase_losses = ForgerLosses.create_from_string(geom_phase_losses)
        self.main_phase_losses = ForgerLosses.create_from_string(main_phase_losses)
        self.stitch_phase_losses = ForgerLosses.create_from_string(stitch_phase_losses)
        self.geom_warmstart_losses = (ForgerLosses.create_from_string(geom_warmstart_losses) if geom_warmstart_losses is not None
                                      else self.geom_phase_losses)
        self.geom_phase_losses.set_partial_loss_with_triband_input(partial_loss_with_triband_input)
        self.main_phase_losses.set_partial_loss_with_triband_input(partial_loss_with_triband_input)
        self.stitcher = stitcher if stitcher is not None else RandomStitcher()
        self.real_sign_sum, self.real_sign_count = 0.0, 0          # 'Loss/signs/real': what ADA adapts p on
        self.pl_batch_shrink, self.pl_decay, self.pl_weight = pl_batch_shrink, pl_decay, pl_weight
        self.pl_mean = torch.zeros([], device=next(G.parameters()).device)

    def G(self, z, c, geom_feature, **kw):
        """``ForgerLoss.run_G``: random noise (the layers' default) + style mixing."""
        return self._G(z, c, geom_feature, noise_mode=self.noise_mode, style_mixing_prob=self.style_mixing_prob, **kw)

    @staticmethod
    def _unwrap(m):
        return getattr(m, "module", m)                                     # DistributedDataParallel wraps the network

    def D(self, img, c=None, sub_batches: int = 1):
        """``ForgerLoss.run_D`` (loss_modified.py:102-107): the discriminator sees augmented images."""
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        return self._D(img, c) if sub_batches == 1 else self._D(img, c, sub_batches=sub_batches)

    def ada_update(self, ada_target: float, batch_size: int, ada_interval: int, ada_kimg: float) -> float:
        """The training loop's ADA heuristic (training_loop_modified.py: adjust p by the sign of E[sign(D(real))] - target)."""
        if self.augment_pipe is None or self.real_sign_count == 0:
            return 0.0
        sign = float(self.real_sign_sum) / self.real_sign_count
        adjust = float(np.sign(sign - ada_target)) * (batch_size * ada_interval) / (ada_kimg * 1000)
        self.augment_pipe.p.copy_((self.augment_pipe.p + adjust).clamp(min=0))
        self.real_sign_sum, self.real_sign_count = 0.0, 0
        return float(self.augment_pipe.p)

    @staticmethod
    def all_reduce_gradients(module, group=None) -> int:
        """Data parallelism of the training step (BASELINE config 5): average the accumulated gradients of ``module`` over the
        ranks with ONE all-reduce of the flattened gradients (RCCL on GPUs; ~8 MB per network), before the optimiser step.
        The reference gets the same average from DistributedDataParallel around each network
        (training_loop_modified.py:243-252, ``misc.ddp_sync``); the explicit form does not depend on which parameters a phase
        touches (the path-length phase differentiates w.r.t. an intermediate tensor, frozen networks take no gradient).
        Returns the number of gradient elements reduced."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and os.environ.get("NB_FORCE_PG") != "1"):
            return 0
        params = [p for p in module.parameters() if p.requires_grad]
        for p in params:                                  # every rank must contribute the same tensors
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.flatten() for p in params])
        dist.all_reduce(flat, group=group)
        flat /= dist.get_world_size(group)
        torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)      # as the reference loop does before a step
        off = 0
        for p in params:
            p.grad.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return off

    def requires_frozen_generator(self) -> bool:
        return self.geom_phase_losses.require_original_fake_image() or self.geom_warmstart_losses.require_original_fake_image()

    def accumulate_gradients(self, phase: str, real_img, geom_feature, gen_z, gain: float = 1.0, positions=None,
                             pl_noise=None, real_geom=None, G_orig=None) -> Dict[str, float]:
        assert phase in ("Gmain", "Greg", "Ggeom", "Ggeom-warm", "Dmain", "Dreg", "Dall")
        g_phase = phase in ("Gmain", "Greg", "Ggeom", "Ggeom-warm")
        self._unwrap(self._G).requires_grad_(g_phase)
        self._unwrap(self._D).requires_grad_(not g_phase)
        try:
            return self._accumulate(phase, real_img, geom_feature, gen_z, gain, positions, pl_noise, real_geom, G_orig)
        finally:
            self._unwrap(self._G).requires_grad_(True)
            self._unwrap(self._D).requires_grad_(True)

    def accumulate_gradients_stitch(self, geom_feature1, geom_feature2, crop1, crop2, gen_z, gain: float = 1.0,
                                    positions1=None) -> Dict[str, float]:
        """``Gstitch`` (loss_modified.py:108-138; the loop runs it every ``stitch_interval`` iterations,
        training_loop_modified.py:264-301): the generator paints two overlapping crops of one drawing, each result is
        composited into the other where they overlap, D judges plain and composite images, and ``stitch_phase_losses``
        (e.g. ``gan(fake_composite)+l1(patch)``) is back-propagated into G."""
        assert not self.stitch_phase_losses.is_empty()
        self._unwrap(self._G).requires_grad_(True)
        self._unwrap(self._D).requires_grad_(False)
        try:
            res = self.stitcher.generate_with_stitching(self._G, gen_z, None, geom_feature1, geom_feature2, crop1, crop2,
                                                        positions1=positions1, noise_mode=self.noise_mode)
            fake = torch.cat([res["fake1"], res["fake2"]], dim=0)
            composite = torch.cat([res["fake1_composite"], res["fake2_composite"]], dim=0)
            fake_logits, composite_logits = self.D(fake, None), self.D(composite, None)
            data = {"fake": fake, "fake_logits": fake_logits, "fake_composite": composite, "fake_composite_logits": composite_logits,
                    "patch1": res["patch1"], "patch2": res["patch2"]}
            loss, vals = self.stitch_phase_losses.compute(data, None)
            loss.mul(gain).backward()
            stats = LazyStats({f"Loss/forger/Gstitch/{k}": v.detach() for k, v in vals.items()})
            stats["Loss/forger/Gstitch/total"] = loss.detach()
            return stats
        finally:
            self._unwrap(self._D).requires_grad_(True)

    def _accumulate(self, phase, real_img, geom_feature, gen_z, gain, positions, pl_noise, real_geom=None, G_orig=None) -> Dict[str, float]:
        stats = LazyStats()
        softplus = torch.nn.functional.softplus
        if phase in ("Ggeom", "Ggeom-warm"):                              # loss_modified.py:181-203
            losses = self.geom_warmstart_losses if phase == "Ggeom-warm" else self.geom_phase_losses
            if not losses.is_empty():
                frozen = losses.require_original_fake_image()
                kw = dict(positions=positions, return_debug_data=True, noise_mode=self.noise_mode)
                gen_img, data = self._G(gen_z, None, geom_feature, style_mixing_prob=0 if frozen else self.style_mixing_prob, **kw)
                data = dict(data)
                data["fake_img"] = gen_img
                if frozen:
                    with torch.no_grad():
                        data["fake_orig"] = G_orig(gen_z, None, geom_feature, positions=positions, style_mixing_prob=0,
                                                   noise_mode=self.noise_mode)
                loss, vals = losses.compute(data, real_geom)
                loss.mean().backward()                                    # (no gain: loss_modified.py:203)
                stats.update({f"Loss/forger/{phase}/{k}": v.detach() for k, v in vals.items()})
        if phase == "Greg" and self.pl_weight != 0:                       # path-length regularisation, loss_modified.py:205-221
            b = max(1, gen_z.shape[0] // self.pl_batch_shrink)
            gen_img, data = self.G(gen_z[:b], None, [g[:b] for g in geom_feature],
                                   positions=None if positions is None else positions[:b], return_debug_data=True)
            if pl_noise is None:
                pl_noise = torch.randn_like(gen_img) / math.sqrt(gen_img.shape[2] * gen_img.shape[3])
            pl_grads, = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[data["ws"]], create_graph=True, only_inputs=True)
            pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
            pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
            self.pl_mean.copy_(pl_mean.detach())
            pl_penalty = (pl_lengths - pl_mean).square()
            (gen_img[:, 0, 0, 0] * 0 + pl_penalty * self.pl_weight).mean().mul(gain).backward()
            stats["Loss/pl_penalty"] = pl_penalty.mean().detach()
        if phase == "Gmain":                                              # maximise logits of generated images
            gen_img, gen_data = self.G(gen_z, None, geom_feature, positions=positions, return_debug_data=True)
            loss = softplus(-self.D(gen_img, None))
            stats["Loss/G/loss"] = loss.mean().detach()
            if not self.main_phase_losses.is_empty():                     # loss_modified.py:170-175
                extra, vals = self.main_phase_losses.compute(gen_data, real_geom)
                loss = loss + extra
                stats.update({f"Loss/forger/Gmain/{k}": v.detach() for k, v in vals.items()})
            loss.mean().mul(gain).backward()
        # (the stacked pass splits the batch in HALVES for the minibatch-stddev statistics: only with as many real as generated
        #  images are those the two passes' groups -- otherwise the two-pass branch below)
        if phase == "Dmain" and self.merge_d_passes and real_img.shape[0] == gen_z.shape[0]:
            # Generated and real images through the discriminator as ONE stacked batch (the reference runs two passes,
            # loss_modified.py:223-238; gradients accumulate linearly, the augmentation draws its parameters per sample and the
            # minibatch-stddev statistics are taken per half, so the result is the same -- at half the launches)
            with torch.no_grad():
                gen_img = self.G(gen_z, None, geom_feature, positions=positions)
            n_gen = gen_img.shape[0]
            logits = self.D(torch.cat([gen_img, real_img.detach().to(gen_img.dtype)], dim=0), None, sub_batches=2)
            gen_logits, real_logits = logits[:n_gen], logits[n_gen:]
            loss_gen, loss_real = softplus(gen_logits), softplus(-real_logits)
            self.real_sign_sum = self.real_sign_sum + real_logits.detach().sign().sum()
            self.real_sign_count += real_logits.numel()
            (loss_gen.mean() + loss_real.mean()).mul(gain).backward()
            stats["Loss/D/loss_gen"] = loss_gen.mean().detach()
            stats["Loss/D/loss_real"] = loss_real.mean().detach()
            return stats
        if phase in ("Dmain", "Dall"):                                    # minimise logits of generated images
            with torch.no_grad():
                gen_img = self.G(gen_z, None, geom_feature, positions=positions)
            loss_gen = softplus(self.D(gen_img, None))
            loss_gen.mean().mul(gain).backward()
            stats["Loss/D/loss_gen"] = loss_gen.mean().detach()
        if phase in ("Dmain", "Dreg", "Dall"):                            # maximise logits of real images (+ R1)
            do_main, do_r1 = phase in ("Dmain", "Dall"), phase in ("Dreg", "Dall") and self.r1_gamma != 0
            real = real_img.detach().requires_grad_(do_r1)
            real_logits = self.D(real, None)
            self.real_sign_sum = self.real_sign_sum + real_logits.detach().sign().sum()      # (device tensor: no sync here)
            self.real_sign_count += real_logits.numel()
            total = real_logits * 0
            if do_main:
                loss_real = softplus(-real_logits)
                total = total + loss_real
                stats["Loss/D/loss_real"] = loss_real.mean().detach()
            if do_r1:
                r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real], create_graph=True, only_inputs=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                total = total + (r1_penalty * (self.r1_gamma / 2))[:, None]
                stats["Loss/r1_penalty"] = r1_penalty.mean().detach()
            total.mean().mul(gain).backward()
        return stats
