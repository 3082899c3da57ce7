"""MI355X-native NeuBE generator behind the reference's ``Generator`` API.

Drop-in for the object the reference stores as ``engine.G`` (``forger/ui/brush.py:631-633``):
same attributes (``z_dim c_dim w_dim img_resolution img_channels num_ws mapping synthesis``,
``synthesis.block_resolutions``, ``synthesis.b{res}.conv{0,1}.noise_const`` ...), the same
keyword-exact call signatures and return conventions as
``thirdparty/stylegan2_ada_pytorch/training/networks_modified.py:123-124, 346-348, 367-368`` and the
same ``state_dict()`` key names, so ``TriadGanPaintEngine._render_stroke_torch`` (brush.py:731-805),
``StyleUVSMapper`` (forger/ui/mapper.py:80-92) and ``PaintStrokeGenerator`` (forger/metrics/util.py)
can call it unmodified.

All arithmetic runs in the hand-written gfx950 kernels of ``csrc/`` through the C ABI of
``include/neube_hip.h``; PyTorch only owns device memory and the stream.  Per forward pass:
1 mapping launch, 1 launch for every layer's affine + demodulation coefficients, 1 launch for every
layer's (position-shifted) constant noise, then one fused modulated-conv launch per layer
(13 at R=128, 15 at R=256) and one fused ToRGB/softmax/triad launch.  (The reference issues ~100
launches per patch: SURVEY 3.3.)
"""
from __future__ import annotations

import ctypes
import dataclasses
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops
from .config import GeneratorConfig, LayerSpec
from .weights import StateDict, random_state_dict, validate_state_dict

_p = ops._p


def _assert_shape(t: torch.Tensor, ref_shape) -> None:
    """``torch_utils/misc.py:80-93`` assert_shape: AssertionError on mismatch, None = wildcard."""
    if t.ndim != len(ref_shape):
        raise AssertionError(f"Wrong number of dimensions: got {t.ndim}, expected {len(ref_shape)}")
    for idx, (size, ref) in enumerate(zip(t.shape, ref_shape)):
        if ref is not None and size != ref:
            raise AssertionError(f"Wrong size for dimension {idx}: got {size}, expected {ref}")


# the arithmetic mode of the conv layers unless a caller asks otherwise (Generator(conv_mode=...)); bench.py times this one
DEFAULT_CONV_MODE = "f8"
# "f16" (round 6): the f8 mode with the correction products of the four large launches skipped -- a plain single-f16 evaluation there
# (1.5e-3 ... 3e-3 from fp32: OUTSIDE the 1e-3 parity budget; the reference's own shipped arithmetic for blocks >= 32^2,
# training/networks.py:634-638).  A timing data point ("what does the split scheme cost"), not a parity mode.
CONV_MODES = ("h3", "f8", "f6", "f32", "f16")
_SPLIT_MODES = ("h3", "f8", "f6", "f16")          # modes whose large layers run on the split-f16 kernel family
_F8_MODES = ("f8", "f6", "f16")                   # ... with f8 operand containers where the channel counts allow

class FullyConnectedLayer(torch.nn.Module):
    """Parameter holder for ``networks.py:92-122``; evaluated inside nb_mapping_f32 / nb_styles_f32."""

    def __init__(self, in_features, out_features, lr_multiplier=1.0):
        super().__init__()
        self.register_buffer("weight", torch.zeros([out_features, in_features]))
        self.register_buffer("bias", torch.zeros([out_features]))
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier


class MappingNetwork(torch.nn.Module):
    """``networks.py:214-290`` for c_dim = 0 (eval mode)."""

    def __init__(self, cfg: GeneratorConfig):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = cfg.z_dim, cfg.c_dim, cfg.w_dim, cfg.num_ws
        self.num_layers = cfg.mapping_layers
        self.lr_multiplier = cfg.mapping_lr_multiplier
        for i in range(self.num_layers):
            setattr(self, f"fc{i}", FullyConnectedLayer(cfg.z_dim if i == 0 else cfg.w_dim, cfg.w_dim,
                                                        cfg.mapping_lr_multiplier))
        self.register_buffer("w_avg", torch.zeros([cfg.w_dim]))
        self._packed = None

    def _pack(self):
        if self._packed is None:
            w = torch.cat([getattr(self, f"fc{i}").weight.reshape(-1) for i in range(self.num_layers)]).contiguous()
            b = torch.cat([getattr(self, f"fc{i}").bias for i in range(self.num_layers)]).contiguous()
            self._packed = (w, b)
            if w.is_cuda and not torch.cuda.is_current_stream_capturing():       # (see SynthesisNetwork._ensure_packed)
                torch.cuda.current_stream(w.device).synchronize()
        return self._packed

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        _assert_shape(z, [None, self.z_dim])
        w, b = self._pack()
        if z.device != w.device:
            raise RuntimeError(f"z is on {z.device} but the generator is on {w.device}")
        z32 = z.to(torch.float32).contiguous()                      # networks.py:261
        n = z32.shape[0]
        x = torch.empty([n, self.num_ws, self.w_dim], dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):                           # (the kernel writes the broadcast of networks.py:278-280)
            _lib.check(_lib.lib().nb_mapping_ws_f32(_p(z32), _p(w), _p(b), _p(x), n, self.z_dim, self.w_dim, self.num_layers,
                                                    self.lr_multiplier, self.num_ws, ops._stream(x)), "mapping")
        if truncation_psi != 1:                                     # networks.py:283-289 (not used by the engine)
            if truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    """Parameter holder for ``networks.py:302-391``.  The arithmetic is one nb_modconv3x3_f32 launch."""

    def __init__(self, spec: LayerSpec, w_dim: int, conv_clamp):
        super().__init__()
        self.spec = spec
        self.resolution, self.up = spec.block_res, spec.up
        self.conv_clamp = conv_clamp
        self.act_gain = math.sqrt(2)
        self.affine = FullyConnectedLayer(w_dim, spec.in_channels)
        self.register_buffer("weight", torch.zeros([spec.out_channels, spec.in_channels, 3, 3]))
        self.register_buffer("noise_strength", torch.zeros([]))
        self.register_buffer("bias", torch.zeros([spec.out_channels]))
        self.register_buffer("noise_grid", torch.zeros([1, spec.block_res, spec.block_res, 2]))
        self.register_buffer("resample_filter", torch.zeros([4, 4]))
        self.register_buffer("noise_const", torch.zeros([spec.block_res, spec.block_res]))


class ToRGBColorTriadLayer(torch.nn.Module):
    """Parameter holder for ``networks.py:415-485`` (color_w_channels = 0, 'triad')."""

    def __init__(self, in_channels: int, w_dim: int, conv_clamp):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.color_w_channels = 0
        self.color_format = "triad"
        self.affine = FullyConnectedLayer(w_dim, in_channels + 9)
        self.register_buffer("weight", torch.zeros([3, in_channels, 1, 1]))
        self.register_buffer("bias", torch.zeros([3]))
        self.register_buffer("color_bias", torch.zeros([9]))
        self.weight_gain = 1 / math.sqrt(in_channels)


class SynthesisBlock(torch.nn.Module):
    """``networks.py:539-680`` for architecture='orig'."""

    def __init__(self, cfg: GeneratorConfig, res: int, layers: Dict[str, LayerSpec]):
        super().__init__()
        self.resolution = res
        self.is_last = res == cfg.img_resolution
        self.architecture = "orig"
        self.use_fp16 = False
        self.num_conv = 0
        self.num_torgb = 0
        self.register_buffer("resample_filter", torch.zeros([4, 4]))
        if res == 4:
            self.in_channels = 0
            self.register_buffer("const", torch.zeros([cfg.channels(4), 4, 4]))
        else:
            spec = layers[f"synthesis.b{res}.conv0"]
            self.in_channels = spec.in_channels
            self.conv0 = SynthesisLayer(spec, cfg.w_dim, cfg.conv_clamp)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(layers[f"synthesis.b{res}.conv1"], cfg.w_dim, cfg.conv_clamp)
        self.num_conv += 1
        if self.is_last:
            self.torgb = ToRGBColorTriadLayer(cfg.channels(res), cfg.w_dim, cfg.conv_clamp)
            self.num_torgb += 1


class _Plan:
    """Device-side layer table (NbLayerDesc[]) + workspaces for a maximum batch size."""

    def __init__(self, syn: "SynthesisNetwork", n_max: int, device):
        cfg = syn.cfg
        specs = cfg.layers
        self.n_max = n_max
        self.device = device
        c_aff = [s.in_channels for s in specs] + [cfg.channels(cfg.img_resolution) + 9]
        c_out = [s.out_channels for s in specs] + [3]
        self.styles = [torch.empty([n_max, c], dtype=torch.float32, device=device) for c in c_aff]
        self.dcoefs = [torch.empty([n_max, c], dtype=torch.float32, device=device) for c in c_out[:-1]]
        self.noise = [torch.empty([n_max, s.block_res, s.block_res], dtype=torch.float32, device=device) for s in specs]
        self.max_res = max(s.block_res for s in specs)
        self.npos_k = None                       # [n_max, 2] normalised positions for the layers that compute their noise themselves (on demand)
        descs = (_lib.NbLayerDesc * (len(specs) + 1))()
        for i, s in enumerate(specs):
            layer = syn.layer_module(s)
            pk = syn.packed[s.name]
            d = descs[i]
            d.affine_w, d.affine_b = layer.affine.weight.data_ptr(), layer.affine.bias.data_ptr()
            d.wsq = pk["wsq"].data_ptr()
            d.styles, d.dcoefs = self.styles[i].data_ptr(), self.dcoefs[i].data_ptr()
            d.noise_const = layer.noise_const.data_ptr()
            d.noise_lin = pk["noise_lin"].data_ptr()
            d.noise_out = self.noise[i].data_ptr()
            d.noise_strength = layer.noise_strength.data_ptr()
            d.c_aff, d.n_plain, d.c_out, d.w_index, d.res = s.in_channels, 0, s.out_channels, s.w_index, s.block_res
            d.style_scale = 1.0
        t = syn.last_block().torgb
        d = descs[len(specs)]
        d.affine_w, d.affine_b = t.affine.weight.data_ptr(), t.affine.bias.data_ptr()
        d.wsq = 0
        d.styles, d.dcoefs = self.styles[-1].data_ptr(), 0
        d.noise_const = d.noise_lin = d.noise_out = d.noise_strength = 0
        d.c_aff, d.n_plain, d.c_out, d.w_index, d.res = c_aff[-1], 9, 3, cfg.torgb_w_index, 0
        d.style_scale = float(t.weight_gain)
        self.host_descs = descs
        self.pack_stream, self.pack_events = None, {}      # side stream for the early geometry packing of this slot
        self.n_layers = len(specs) + 1
        self.const_rep = None                              # b4.const repeated over the batch (filled at first use)
        self._noise_cut = {}
        self.table = self._upload(descs)

    def _upload(self, descs) -> torch.Tensor:
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        return torch.from_numpy(raw).to(self.device)

    def table_without_noise_from(self, first: int) -> torch.Tensor:
        """The layer table with ``noise_const`` cleared for layers >= ``first`` (they compute their noise inside the
        convolution, NbNoiseSrc): the fused styles + noise launch of small batches then skips their noise images."""
        t = self._noise_cut.get(first)
        if t is None:
            descs = (_lib.NbLayerDesc * self.n_layers)()
            ctypes.memmove(descs, self.host_descs, ctypes.sizeof(descs))
            for i in range(first, self.n_layers - 1):
                descs[i].noise_const = 0
            t = self._noise_cut[first] = self._upload(descs)
        return t

    def table_with_noise_overrides(self, syn, noise_buffers) -> torch.Tensor:
        """Reference noise_buffers (networks_modified.py:163-165): per-call replacement of noise_const."""
        descs = (_lib.NbLayerDesc * self.n_layers)()
        ctypes.memmove(descs, self.host_descs, ctypes.sizeof(descs))
        keep = []
        for i, s in enumerate(syn.cfg.layers):
            key = f"b{s.block_res}.conv{0 if s.up == 2 else 1}.noise_const"
            buf = noise_buffers.get(key)
            if buf is not None:
                if not torch.is_tensor(buf):
                    buf = torch.from_numpy(np.asarray(buf))
                buf = buf.to(device=self.device, dtype=torch.float32).contiguous()
                _assert_shape(buf, [s.block_res, s.block_res])
                keep.append(buf)
                descs[i].noise_const = buf.data_ptr()
        return self._upload(descs), keep


@dataclasses.dataclass
class _PassOptions:
    """The private options of one synthesis pass -- the underscore keywords of ``SynthesisNetwork.forward`` that this build's
    schedules use (the public keywords are the reference's).  One place for their meaning and for the illegal combinations."""
    noise_mode: str = "random"                     # SynthesisLayer default, networks.py:362
    norm_noise_positions: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None       # `_positions`: integer (y, x) patch positions, normalised in-kernel
    extra: Optional[dict] = None                   # `_extra_outputs`: logits / RGBA / uint8 compositing of the fused ToRGB launch
    # split entry of the tiled-canvas schedule (painting.py): `_stop_after=res` returns the output of block `res` before any
    # blending; `_resume=(res, x)` continues after block `res` from (blended) features x
    stop_after: Optional[int] = None
    resume: Optional[tuple] = None
    plan_slot: int = 0                             # `_plan_slot`: workspace to use (passes that may overlap on different streams)
    # `_reuse_styles`: the styles / demodulation coefficients (and the noise images of the small layers) of THIS batch are already
    # in the workspace slot -- an earlier pass of the same batch computed them (pipeline.TriadStepPipeline: the head pass computes
    # every layer's styles, the tail pass of the same step resumes from its features)
    reuse_styles: bool = False
    # `_prepare_only`: enqueue what a pass needs BEFORE its first layer -- every layer's styles and demodulation coefficients, the
    # small layers' noise images (all into the workspace slot) and the early geometry packs -- and return a handle;
    # `_prepared=handle`: the pass of the same batch on the same slot that starts from it (pipeline.TriadPrefetchPipeline runs the
    # former for step k+1 under the last layer of step k).  `_mark=(event, layer name)`: record the event on the current stream
    # right before that layer's launch.
    prepare_only: bool = False
    prepared: Optional[dict] = None
    mark: Optional[tuple] = None

    _KEYS = {"noise_mode": "noise_mode", "norm_noise_positions": "norm_noise_positions", "_positions": "positions",
             "_extra_outputs": "extra", "_stop_after": "stop_after", "_resume": "resume", "_plan_slot": "plan_slot",
             "_reuse_styles": "reuse_styles", "_prepare_only": "prepare_only", "_prepared": "prepared", "_mark": "mark"}

    @classmethod
    def from_kwargs(cls, kw: dict) -> "_PassOptions":
        kw = dict(kw)
        kw.pop("force_fp32", None)       # always fp32 here (SURVEY note B)
        kw.pop("fused_modconv", None)    # one arithmetic form (csrc/nb_modconv.hip)
        o = cls(**{field: kw.pop(key) for key, field in cls._KEYS.items() if key in kw})
        if kw:
            raise TypeError(f"unexpected synthesis kwargs: {sorted(kw)}")
        if o.noise_mode not in ("random", "const", "none"):
            raise AssertionError(f"noise_mode {o.noise_mode!r}")
        return o


@dataclasses.dataclass
class _Pass:
    """State of one synthesis pass: what ``_begin_pass`` validated, what ``_prepare`` enqueued, and the activations as they move
    through ``_run_layers`` (x: fp32 NCHW, x2: the geometry feature still to be concatenated, x_h2: the next layer's complete input
    in operand format when its producer wrote it)."""
    opts: _PassOptions
    ws: torch.Tensor
    n: int
    device: torch.device
    plan: "_Plan"
    lazy_geom: object
    geom_feature: list
    return_debug_data: bool
    return_features: list
    blended_features: dict
    noise_buffers: Optional[dict]
    table: torch.Tensor
    stream: int = 0
    npos: Optional[torch.Tensor] = None
    ipos: Optional[torch.Tensor] = None
    npos_k: Optional[torch.Tensor] = None          # ipos normalised once per batch for the layers that compute their noise themselves
    shared: bool = False                     # constant noise without positions: one image for the whole batch
    inkernel_from: Optional[int] = None      # first layer (resolution order) from which every layer computes its noise itself
    pre_h2: dict = dataclasses.field(default_factory=dict)
    keep_alive: list = dataclasses.field(default_factory=list)
    x: Optional[torch.Tensor] = None
    x2: Optional[torch.Tensor] = None
    x_h2: Optional[torch.Tensor] = None
    geo_idx: int = 0
    packs_waited: bool = False
    fused_rgb: Optional[tuple] = None
    img: Optional[torch.Tensor] = None
    debug_data: dict = dataclasses.field(default_factory=dict)


class SynthesisNetwork(torch.nn.Module):
    """``networks_modified.py:28-223`` on the HIP kernels."""

    def __init__(self, cfg: GeneratorConfig):
        super().__init__()
        self.cfg = cfg
        self.w_dim = cfg.w_dim
        self.img_resolution = cfg.img_resolution
        self.img_resolution_log2 = int(math.log2(cfg.img_resolution))
        self.img_channels = cfg.img_channels
        self.block_resolutions = cfg.block_resolutions
        self.geom_feature_resolutions = list(cfg.geom_feature_resolutions)
        self.geom_feature_channels = list(cfg.geom_feature_channels)
        self.geom_linear = None
        self.pos_encoding_channels = 0
        self.pos_encoding_feature_resolutions = []
        self.pos_encoding_injection_mode = "cat"
        self.num_ws = cfg.num_ws
        layers = {l.name: l for l in cfg.layers}
        for res in self.block_resolutions:
            setattr(self, f"b{res}", SynthesisBlock(cfg, res, layers))
        self.packed: Dict[str, Dict[str, torch.Tensor]] = {}
        self._plans: Dict[int, _Plan] = {}      # per-batch workspaces, one per concurrent sub-batch (slot)
        # "f8" (default): split-f16 products with both correction terms on block-scaled fp8 MFMAs (pixels within 1e-4 of
        # fp32 on O(1) activations, 1.4e-4 with ten layers at the conv_clamp: tests/golden/gen_hdr_r128.npz; budget 1e-3);
        # "h3": all three hi/lo products on the f16 matrix cores (5e-6); "f32": every layer on the exact-fp32 MFMA kernels.
        self.conv_mode = DEFAULT_CONV_MODE
        # a layer takes the split-f16 kernels when it has enough output pixels to fill the chip with their (large)
        # workgroups: below ~64 workgroups the fp32 kernels (smaller tiles, split-K) have the lower latency
        # (tools/layers_b1.py: at batch 1 the >= 128x128 layers gain 25-65 %, the <= 64x64 layers lose 40-100 %)
        self.h3_min_pixels = 128 * 128
        self.h3_up2_w16_min_batch = 16        # 16x16 -> 32x32 conv0 on the large up=2 kernel (8 x 16 quad tiles) from this batch
        self.h3_up2_w8_min_batch = 16         # 8x8 -> 16x16 conv0 likewise (one 8 x 8 quad tile per sample and c_out slice): the
                                              # FIR-folded form on the small-image kernel re-reads 147 KB of weights per 32 positions
        self.h3_min_batch = 1
        # the latency-oriented styles / demodulation launch needs 16-byte friendly shapes (every style1 shape has them)
        self._styles_fast = cfg.w_dim % 16 == 0 and all(l.out_channels % 4 == 0 for l in cfg.layers)
        self._h3_batch_ok = True
        self._n = 1
        self.h2_handoff = True            # split-f16 layers write the next layer's H2 input directly (no pack pass)
        self.early_geom_pack = True       # geometry channels of such inputs are packed at the start, on a side stream
        self.fuse_torgb = True            # last conv1 + ToRGB + compositing in one launch (split-f16 path)
        self.noise_in_kernel = True       # large split-f16 layers compute their (position-shifted) noise themselves
        self.positions_once = True        # ... from positions normalised ONCE per batch (nb_norm_positions_f32) instead of per tile (batches > 8)
        self.layer_kernels: Dict[str, str] = {}
        self.layer_formats: Dict[str, int] = {}      # operand format each split-f16 layer last ran with (0 H2, 1 f8, 2 f6)

    # -- helpers --
    def get_last_block(self):
        return getattr(self, f"b{self.block_resolutions[-1]}")

    last_block = get_last_block

    def layer_module(self, spec: LayerSpec) -> SynthesisLayer:
        return getattr(getattr(self, f"b{spec.block_res}"), "conv0" if spec.up == 2 else "conv1")

    def invalidate(self):
        self.packed = {}
        self._plans = {}

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def _ensure_packed(self):
        if self.packed:
            return
        for s in self.cfg.layers:
            layer = self.layer_module(s)
            wpk, wsq = ops.pack_conv_weight(layer.weight)
            self.packed[s.name] = {"wpk": wpk, "wsq": wsq,
                                   "noise_lin": layer.noise_grid[0, :, 0, 0].contiguous(),
                                   # transposed copy for the convolutions that compute their noise themselves (NbNoiseSrc)
                                   "noise_const_t": layer.noise_const.t().contiguous()}
            if self.conv_mode in _SPLIT_MODES and self.cfg.conv_clamp is not None:
                self.packed[s.name]["w_h3"] = ops.pack_conv_weight_h3(layer.weight)
                if self.conv_mode in _F8_MODES and s.in_channels % 16 == 0:
                    self.packed[s.name]["w_f8"] = ops.pack_conv_weight_h3f8(layer.weight)
                    if self.conv_mode == "f6" and s.up == 2:
                        self.packed[s.name]["w_f6"] = ops.pack_conv_weight_h3f6(layer.weight)
                if s.up == 2 and s.in_res <= 32 and s.in_channels % 16 == 0:
                    self.packed[s.name]["w_h3_up2"] = ops.pack_conv_weight_h3_up2_phases(layer.weight, layer.resample_filter)
        t = self.get_last_block().torgb
        self.packed["torgb"] = {"w": t.weight.reshape(3, -1).contiguous()}
        # The packed tensors were written by launches on the CURRENT stream, but every stream uses them from now on (the tiled
        # schedule alternates its batches over side streams, and the first use after a mode switch / weight load may well sit on
        # one of them): without this, a batch on another stream could read a weight block that is still being written -- seen as
        # a few wrong pixels in ~1 of 5 canvases painted right after set_conv_mode('f32').  Packing is rare; one sync is cheap.
        if t.weight.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(t.weight.device).synchronize()

    def _h3_eligible(self, s: LayerSpec) -> bool:
        """conv1 layers that run as 3-pass split-f16 MFMA (csrc/nb_modconv_h3.hip): the kernel needs rows of 32
        pixels and 16-row tiles, and a conv_clamp so that activations are bounded inside the f16 range."""
        return (self.conv_mode in _SPLIT_MODES and self._h3_batch_ok and s.up == 1 and s.block_res >= 32
                and self._n * s.block_res ** 2 >= self.h3_min_pixels and s.block_res % 32 == 0 and self.cfg.conv_clamp is not None and self.cfg.conv_clamp <= 1024)

    def _h3_up2_eligible(self, s: LayerSpec) -> bool:
        """conv0 (up=2) layers that run on the split-f16 4-phase kernel: input rows must be multiples of 32 pixels, or 16
        pixels (8 x 16 quad tiles: two workgroups per sample and c_out slice, so only worth it at batch >= 16)."""
        w8 = s.in_res == 8 and self._n >= self.h3_up2_w8_min_batch
        return (self.conv_mode in _SPLIT_MODES and self._h3_batch_ok and s.up == 2
                and ((s.in_res >= 32 and s.in_res % 32 == 0) or (s.in_res == 16 and self._n >= self.h3_up2_w16_min_batch) or w8)
                and (w8 or self._n * s.block_res ** 2 >= self.h3_min_pixels)
                and self.cfg.conv_clamp is not None and self.cfg.conv_clamp <= 1024)

    small_h3 = True        # <= 64x64 conv1 layers on the small-tile split-f16 kernel (csrc/nb_modconv_small.hip)

    def _small_h3_eligible(self, s: LayerSpec) -> bool:
        """conv1 layers too small for the large-tile split-f16 kernel: same hi/lo products on 32 x 32 tiles with K split
        over the waves, fp32 in and out (needs whole 16-channel chunks and the conv_clamp bound like the other f16 paths)."""
        return (self.small_h3 and self.conv_mode in _SPLIT_MODES and s.up == 1 and s.block_res <= 64
                and s.in_channels % 16 == 0 and s.in_channels <= 512
                and self.cfg.conv_clamp is not None and self.cfg.conv_clamp <= 1024)

    def _small_h3_up2_eligible(self, s: LayerSpec) -> bool:
        """conv0 (up = 2) layers with inputs <= 32x32 that the large-tile up=2 kernel does not take."""
        return (self.small_h3 and self.conv_mode in _SPLIT_MODES and s.up == 2 and s.in_res <= 32
                and s.in_channels % 16 == 0 and s.in_channels <= 512 and s.name in self.packed
                and "w_h3_up2" in self.packed[s.name]
                and self.cfg.conv_clamp is not None and self.cfg.conv_clamp <= 1024)

    def _operand_fmt(self, s: Optional[LayerSpec]) -> int:
        """Operand format of a split-f16 layer's input: 1 = "f8" (correction products on block-scaled fp8 MFMAs; whole
        16-channel chunks only), 0 = H2 (hi/lo f16); 2 = "f6" (round 5: fp6 correction products with per-pixel block scales) -- in
        conv_mode "f6", for the layers whose kernel takes it: the up=2 launches that run on the 12-row software-pipelined kernel
        (their producers, an up=1 kernel and the geometry pack, write the format; the up=2 epilogue does not, so the up=1 layers
        behind an up=2 layer stay on f8 operands)."""
        if s is None or self.conv_mode not in _F8_MODES or s.in_channels % 16:
            return 0
        if self.conv_mode == "f6" and s.up == 2 and s.in_res % 32 == 0 and self._up2_h3_variant_name(2, self._n, s) == "modconv3x3_up2v_kernel":
            return 2
        return 1

    def _variant_name(self, n: int, s: LayerSpec) -> str:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().nb_modconv3x3_variant(n, s.in_res, s.in_res, s.out_channels, s.up, buf, 128), "variant")
        return buf.value.decode()

    def _up2_h3_variant_name(self, in_fmt: int, n: int, s: LayerSpec) -> str:
        """Which up=2 kernel the library picks for this problem (a label for layer_kernels; resolved once per problem)."""
        key = (in_fmt, n, s.name)
        cache = self.__dict__.setdefault("_up2_variant_names", {})
        if key not in cache:
            buf = ctypes.create_string_buffer(128)
            _lib.check(_lib.lib().nb_modconv3x3_up2_h3_variant(in_fmt, s.in_channels, s.out_channels, n, s.in_res, s.in_res, buf, 128), "variant")
            cache[key] = buf.value.decode()
        return cache[key]

    def _get_plan(self, n: int, device, slot: int = 0) -> _Plan:
        self._ensure_packed()
        plan = self._plans.get(slot)
        if plan is None or plan.n_max < n or plan.device != device:
            if plan is not None and plan.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                # the old workspace may still be read by kernels on ANOTHER stream than the one it returns to when freed
                # (a slot is used from side streams): let them finish before it goes back to the allocator (rare: growth)
                torch.cuda.synchronize(plan.device)
            plan = _Plan(self, max(n, 1 if plan is None else plan.n_max), device)
            self._plans[slot] = plan
        return plan

    # -- optional per-launch HIP-event timing (bench.py): events are recorded on the launch stream --
    layer_events = None      # set to a list to collect (name, start_event, end_event)
    event_filter = None      # optional set of names: record only these (every recorded pair costs ~10 us of stream time)
    event_pool = None        # optional list of pre-created (and once-recorded) timing events to draw from: creating
                             # events inside a timed region costs milliseconds per step in a fresh process

    def _begin_event(self, name):
        if self.layer_events is None or (self.event_filter is not None and name not in self.event_filter):
            return None
        if self.event_pool:
            e0, e1 = self.event_pool.pop(), self.event_pool.pop()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0, e1)

    def _end_event(self, ev):
        if ev is not None:
            ev[2].record()
            self.layer_events.append(ev)

    # -- forward --
    def forward(self, ws, geom_feature, pos_encoding=None, return_debug_data=False, return_features=None,
                blended_features=None, noise_buffers=None, **block_kwargs):
        """``networks_modified.py:123-223``.  Public keywords as the reference; the underscore keywords are the private options of
        this build's schedules (``_PassOptions``).  Three steps: ``_begin_pass`` (validation, workspace), ``_prepare`` (styles,
        noise images, geometry operands: everything a pass needs before its first layer), ``_run_layers``."""
        opts = _PassOptions.from_kwargs(block_kwargs)
        if pos_encoding is not None:
            raise RuntimeError("positional encodings are not part of the shipped configuration (SG/train.py:680)")
        ps = self._begin_pass(ws, geom_feature, opts, return_debug_data, return_features, blended_features, noise_buffers)
        with torch.cuda.device(ps.device):
            ps.stream = ops._stream(ps.ws)
            self._prepare(ps)
            if opts.prepare_only:
                return {"pre_h2": ps.pre_h2, "keep": ps.keep_alive, "n": ps.n, "slot": opts.plan_slot, "plan": ps.plan}
            return self._run_layers(ps)

    def _begin_pass(self, ws, geom_feature, opts: "_PassOptions", return_debug_data, return_features, blended_features,
                    noise_buffers) -> "_Pass":
        """Validate the call (networks_modified.py:145), pick the workspace slot and collect the state of one pass."""
        _assert_shape(ws, [None, self.num_ws, self.w_dim])          # networks_modified.py:145
        device = self.get_last_block().conv1.weight.device
        if ws.device != device:
            raise RuntimeError(f"ws is on {ws.device} but the generator is on {device}")
        ws = ws.to(torch.float32).contiguous()
        n = ws.shape[0]
        self._h3_batch_ok = n >= self.h3_min_batch
        self._n = n
        plan = self._get_plan(n, device, opts.plan_slot)
        lazy_geom = geom_feature if hasattr(geom_feature, "encode_for") else None      # encoder.LazyGeometry
        return_features = [] if return_features is None else return_features
        blended_features = {} if blended_features is None else blended_features
        if opts.prepare_only or opts.prepared is not None:
            if (lazy_geom is not None or opts.noise_mode != "const" or noise_buffers or opts.resume is not None
                    or opts.stop_after is not None or opts.reuse_styles or blended_features):
                raise RuntimeError("_prepare_only / _prepared: whole passes with constant noise and plain geometry tensors only")
            h = opts.prepared
            if h is not None and (h["n"] != n or h["slot"] != opts.plan_slot or h["plan"] is not plan):
                raise RuntimeError("_prepared: the handle belongs to another batch size or workspace slot")
        if lazy_geom is None:
            geom_feature = list(geom_feature) if isinstance(geom_feature, (list, tuple)) else [geom_feature]
        return _Pass(opts=opts, ws=ws, n=n, device=device, plan=plan, lazy_geom=lazy_geom, geom_feature=geom_feature,
                     return_debug_data=return_debug_data, return_features=return_features, blended_features=blended_features,
                     noise_buffers=noise_buffers, table=plan.table)

    # ---- step 1: what a pass needs before its first layer ----
    def _prepare(self, ps: "_Pass") -> None:
        self._prepare_noise_sources(ps)
        self._launch_styles_and_noise(ps)
        # Geometry channels of the layers that receive their input in H2 / f8 operand format: packed NOW on a side
        # stream (they only need the consumer's styles), under the small first layers, instead of between the
        # producer and the consumer.  pre_h2[res] = (consumer input tensor, event).
        ps.pre_h2 = {} if ps.opts.prepared is None else dict(ps.opts.prepared["pre_h2"])
        if ps.lazy_geom is not None:
            self._encode_lazy_geometry(ps)
        if self.early_geom_pack and self.h2_handoff and ps.opts.prepared is None:
            self._pack_geometry_early(ps)

    def _prepare_noise_sources(self, ps: "_Pass") -> None:
        """Per-call noise overrides, the patch positions in the form the kernels take them, and the first layer from which every
        layer computes its (position-shifted) noise itself."""
        cfg, opts, plan, n, device = self.cfg, ps.opts, ps.plan, ps.n, ps.device
        if opts.noise_mode == "const":
            if ps.noise_buffers:
                ps.table, keep = plan.table_with_noise_overrides(self, ps.noise_buffers)
                ps.keep_alive += [ps.table] + keep
            if opts.positions is not None:
                ps.ipos = opts.positions.to(device=device, dtype=torch.int64).contiguous()
                _assert_shape(ps.ipos, [n, 2])
                ps.keep_alive.append(ps.ipos)
            elif opts.norm_noise_positions is not None:
                ps.npos = opts.norm_noise_positions.to(device=device, dtype=torch.float32).contiguous()
                _assert_shape(ps.npos, [n, 2])
                ps.keep_alive.append(ps.npos)
            ps.shared = ps.npos is None and ps.ipos is None
            if self.positions_once and ps.ipos is not None and n > 8:
                # Integer positions -> normalised ones ONCE per batch (nb_norm_positions_f32: the function the kernels use, same bits).  The layers
                # that compute their noise themselves normalise at the top of EVERY tile -- from the integers four 64-bit modulo operations
                # per lane and tile, 44 us of a 1.93 ms step at batch 32 --, the noise launch once per block.  (Batches <= 8 keep the
                # integers: few tiles, and a launch costs more than it would save.)
                if plan.npos_k is None:
                    plan.npos_k = torch.empty([plan.n_max, 2], dtype=torch.float32, device=device)
                _lib.check(_lib.lib().nb_norm_positions_f32(_p(ps.ipos), self.img_resolution, _p(plan.npos_k), n, ps.stream), "norm_positions")
                ps.npos_k = plan.npos_k
        # first layer (in resolution order) from which every layer runs on the large split-f16 kernels: those compute their
        # position-shifted noise in their own prologue.  Not with per-call noise buffers (their transposes do not exist).
        if (self.noise_in_kernel and opts.noise_mode == "const" and ps.table is plan.table and (ps.npos is not None or ps.ipos is not None)):
            elig = [(self._h3_up2_eligible(sp) if sp.up == 2 else self._h3_eligible(sp)) for sp in cfg.layers]
            k_ = len(elig)
            while k_ > 0 and elig[k_ - 1]:
                k_ -= 1
            ps.inkernel_from = k_ if k_ < len(elig) else None

    def _launch_styles_and_noise(self, ps: "_Pass") -> None:
        """Every layer's affine + demodulation coefficients (one launch) and the small layers' noise images (one launch), into the
        workspace slot -- unless an earlier pass of the same batch left them there."""
        cfg, opts, plan, n, lib, stream = self.cfg, ps.opts, ps.plan, ps.n, _lib.lib(), ps.stream
        table, npos, ipos, inkernel_from, noise_mode = ps.table, ps.npos, ps.ipos, ps.inkernel_from, opts.noise_mode
        if ps.npos_k is not None:
            npos, ipos = ps.npos_k, None         # (normalised once per batch, see _prepare_noise_sources)
        resume, stop_after = opts.resume, opts.stop_after
        if opts.prepared is not None:
            return                                  # styles, coefficients and noise images are in the workspace slot already
        if opts.reuse_styles:
            if resume is None or (inkernel_from is None and noise_mode == "const") or table is not plan.table:
                raise RuntimeError("_reuse_styles needs a resumed pass whose layers compute their noise themselves")
            if any(i_ < inkernel_from for i_, sp in enumerate(cfg.layers) if sp.block_res > resume[0]) and noise_mode == "const":
                raise RuntimeError("_reuse_styles: a resumed layer would need a noise image that no launch of this pass writes")
            return
        if (self._styles_fast and noise_mode == "const" and table is plan.table and n <= 8 and (npos is not None or ipos is not None)):
            # small batches: styles + per-sample noise in one launch (a launch costs more than either computes)
            tbl = plan.table if inkernel_from is None else plan.table_without_noise_from(inkernel_from)
            _lib.check(lib.nb_styles_noise_f32(_p(tbl), plan.n_layers, _p(ps.ws), self.num_ws, self.w_dim, _p(npos),
                                               _p(ipos), self.img_resolution, n, stream), "styles_noise")
            return
        styles_fn = lib.nb_styles_fast_f32 if self._styles_fast else lib.nb_styles_f32
        _lib.check(styles_fn(_p(plan.table), plan.n_layers, _p(ps.ws), self.num_ws, self.w_dim, n, stream), "styles")
        if noise_mode == "const":
            # only the layers this pass runs (the tiled-canvas schedule splits the generator at R/2: the head pass
            # needs no 256x256 noise images, the tail pass nothing but those); layers are ordered by resolution
            lo_ = 0 if resume is None else sum(1 for sp in cfg.layers if sp.block_res <= resume[0])
            hi_ = plan.n_layers if stop_after is None else sum(1 for sp in cfg.layers if sp.block_res <= stop_after)
            # layers on the large split-f16 kernels compute their shifted noise themselves (NbNoiseSrc): the noise
            # launch stops at the first of them (layers are ordered by resolution, eligibility grows with it)
            if inkernel_from is not None:
                hi_ = min(hi_, inkernel_from)
            if hi_ > lo_:
                _lib.check(lib.nb_noise_f32(table.data_ptr() + lo_ * ctypes.sizeof(_lib.NbLayerDesc), hi_ - lo_,
                                            max(sp.block_res for sp in cfg.layers[lo_:hi_]), _p(npos), _p(ipos),
                                            self.img_resolution, n, stream), "noise")

    def _geometry_consumers(self, ps: "_Pass"):
        """(g_idx, gres, producer index / spec, consumer index / spec, geometry channels, consumer operand format) of every geometry
        feature that could be handed to its consumer in operand format: injected below the image resolution, in a block this pass
        runs, and neither tapped nor blended at its resolution."""
        cfg, opts = self.cfg, ps.opts
        specs_ = {s_.name: (i_, s_) for i_, s_ in enumerate(cfg.layers)}
        for g_idx, gres in enumerate(self.geom_feature_resolutions):
            if (opts.resume is not None and gres <= opts.resume[0]) or gres >= cfg.img_resolution:
                continue
            if gres in ps.return_features or gres in ps.blended_features or opts.stop_after == gres:
                continue
            ip, sp_ = specs_[f"synthesis.b{gres}.conv1"]
            ic, sc_ = specs_[f"synthesis.b{2 * gres}.conv0"]
            yield g_idx, gres, ip, sp_, ic, sc_, self.geom_feature_channels[g_idx], self._operand_fmt(sc_)

    def _encode_lazy_geometry(self, ps: "_Pass") -> None:
        """Geometry not encoded yet (encoder.LazyGeometry): let the encoder write the features that feed an H2 / f8 layer input
        straight into that layer's operand tensor (x the consumer's styles, which exist now); the rest comes back as fp32."""
        opts, plan, n, device, lazy_geom = ps.opts, ps.plan, ps.n, ps.device, ps.lazy_geom
        targets = {}
        if self.h2_handoff:
            for g_idx, gres, ip, sp_, ic, sc_, gch, ofmt in self._geometry_consumers(ps):
                if g_idx != 1:
                    continue                        # (feature 0 also feeds the encoder's own decoder: it stays fp32)
                # (the encoder's hand-off epilogue writes H2 / f8 operands only: an f6 consumer gets the feature back in
                #  fp32 and the pack path writes its operands)
                if (ofmt != 2 and self._h3_eligible(sp_) and self._h3_up2_eligible(sc_) and sp_.out_channels % 16 == 0 and gch % 16 == 0
                        and tuple(lazy_geom.feature_shape(g_idx)) == (n, gch, gres, gres)
                        and getattr(lazy_geom, "can_handoff", lambda i_: True)(g_idx)):
                    dst = torch.empty(ops.h2_shape(n, sc_.in_channels, gres, gres), dtype=torch.float16, device=device)
                    c_prod = sc_.in_channels - gch
                    targets[g_idx] = dict(dst=dst, scale_ptr=plan.styles[ic].data_ptr() + 4 * c_prod, scale_stride=sc_.in_channels,
                                          c8_total=sc_.in_channels // 8, cg0=c_prod // 8, fmt=ofmt)
                    ps.pre_h2[gres] = (dst, None)
        needed = [gres for gres in self.geom_feature_resolutions if opts.resume is None or gres >= opts.resume[0]]
        if not needed:
            ps.geom_feature = [None] * len(self.geom_feature_resolutions)      # a resumed pass past the last injection
        else:
            ps.geom_feature = list(lazy_geom.encode_for(targets))
            if any(ps.geom_feature[k] is None and k not in targets for k in range(len(ps.geom_feature))):
                raise RuntimeError("geometry provider returned no tensor for a feature the generator needs in fp32")

    def _pack_geometry_early(self, ps: "_Pass") -> None:
        """fp32 geometry features x the consumer's styles -> the consumer's operand tensor, on the slot's side stream."""
        plan, n, device, lib, geom_feature = ps.plan, ps.n, ps.device, _lib.lib(), ps.geom_feature
        pack_waited = False           # has plan.pack_stream been ordered behind this call's styles launch yet?
        for g_idx, gres, ip, sp_, ic, sc_, gch, ofmt in self._geometry_consumers(ps):
            if g_idx >= len(geom_feature) or gres in ps.pre_h2 or geom_feature[g_idx] is None:
                continue
            if not (self._h3_eligible(sp_) and self._h3_up2_eligible(sc_) and sp_.out_channels % 8 == 0
                    and (ofmt == 0 or (sp_.out_channels % 16 == 0 and gch % 16 == 0))):
                continue
            g = geom_feature[g_idx]
            if g.device != device or g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != (n, gch, gres, gres):
                continue                                           # (the in-line path validates and converts)
            cur = torch.cuda.current_stream(device)
            if plan.pack_stream is None:
                plan.pack_stream = torch.cuda.Stream(device=device)
                plan.pack_events = {}
            dst = torch.empty(ops.h2_shape(n, sc_.in_channels, gres, gres), dtype=torch.float16, device=device)
            # (Measured, round 3: this memory-bound pack -- 201 MB per batch of 32 at R=256 -- costs the kernels it runs
            #  beside ~45 us wherever it is placed: here, beside the small first layers (b8.conv1 12 -> 58 us), or deferred to
            #  the b32 / b64 layers (+30 / +33 us), with its grid capped or not.  The painting engine never runs it: its
            #  encoder writes the operand format directly, encoder.LazyGeometry.)
            if not pack_waited:
                # the styles (and, on the lazy path, the encoder's fp32 features) are enqueued on `cur`; tracked apart
                # from pre_h2, which the lazy path may already have filled for another feature
                plan.pack_stream.wait_stream(cur)
                pack_waited = True
            part = (lib.nb_pack_h2_part_f32, lib.nb_pack_h2f8_part_f32, lib.nb_pack_h2f6_part_f32)[ofmt]
            c_prod = sc_.in_channels - gch
            _lib.check(part(_p(g), gch, plan.styles[ic].data_ptr() + 4 * c_prod, sc_.in_channels, _p(dst),
                            (sc_.in_channels + 7) // 8, c_prod // 8, n, gres * gres, plan.pack_stream.cuda_stream), "pack_h2_part")
            ev_ = plan.pack_events.get(gres)
            if ev_ is None:
                ev_ = plan.pack_events[gres] = torch.cuda.Event()
            ev_.record(plan.pack_stream)
            plan.pack_last_event = ev_
            g.record_stream(plan.pack_stream)
            dst.record_stream(plan.pack_stream)
            ps.pre_h2[gres] = (dst, ev_)

    # ---- step 2: the blocks ----
    def _run_layers(self, ps: "_Pass"):
        cfg, opts, plan, n, device = self.cfg, ps.opts, ps.plan, ps.n, ps.device
        resume = opts.resume
        specs = {s.name: (i, s) for i, s in enumerate(cfg.layers)}
        for res in self.block_resolutions:
            block = getattr(self, f"b{res}")
            if resume is not None and res <= resume[0]:
                if res == resume[0]:
                    ps.x = resume[1].to(device=device, dtype=torch.float32).contiguous()
                    _assert_shape(ps.x, [n, cfg.channels(res), res, res])
                if res in self.geom_feature_resolutions:
                    if res == resume[0]:
                        ps.x2 = ps.geom_feature[ps.geo_idx].to(torch.float32).contiguous()
                    ps.geo_idx += 1
                continue
            names = ([f"synthesis.b{res}.conv0"] if res > 4 else []) + [f"synthesis.b{res}.conv1"]
            if res == 4:
                # networks.py:641-643: the learned constant repeated over the batch -- a constant of the weights, so the
                # workspace keeps the repeated tensor (built once per maximum batch, dropped with the plan on a reload)
                if plan.const_rep is None:
                    plan.const_rep = block.const.unsqueeze(0).expand(plan.n_max, -1, -1, -1).contiguous()
                ps.x = plan.const_rep[:n]
            elif ps.x is not None:                  # (None: the previous block handed its output over in H2 format)
                _assert_shape(ps.x, [None, block.in_channels - (0 if ps.x2 is None else ps.x2.shape[1]), res // 2, res // 2])
            for name in names:
                self._run_layer(ps, block, res, *specs[name])
            if opts.stop_after is not None and res == opts.stop_after:
                return ps.x
            self._finish_block(ps, block, res, specs)
        if len(ps.debug_data) > 0:
            return ps.img, ps.debug_data
        return ps.img

    def _run_layer(self, ps: "_Pass", block, res: int, i: int, s: LayerSpec) -> None:
        """One SynthesisLayer (networks.py:362-391) = one fused launch (+ a pack launch in front of a split-f16 layer whose
        producer was not one).  Reads ps.x / ps.x2 / ps.x_h2, leaves the layer's output there."""
        cfg, opts, plan, n, device, lib, stream = self.cfg, ps.opts, ps.plan, ps.n, ps.device, _lib.lib(), ps.stream
        name, noise_mode, extra = s.name, opts.noise_mode, opts.extra
        x, x2, x_h2 = ps.x, ps.x2, ps.x_h2
        return_features, blended_features, stop_after = ps.return_features, ps.blended_features, opts.stop_after
        layer = self.layer_module(s)
        pk = self.packed[name]
        if opts.mark is not None and name == opts.mark[1]:
            opts.mark[0].record(torch.cuda.current_stream(device))
        c2 = 0 if x2 is None else x2.shape[1]
        c1 = s.in_channels - c2 if x is None else x.shape[1]
        if c1 + c2 != s.in_channels:
            raise AssertionError(f"{name}: got {c1}+{c2} input channels, expected {s.in_channels}")
        noise_ptr, nstride = None, 0
        if noise_mode == "const" and ps.inkernel_from is not None and i >= ps.inkernel_from:
            nsrc = _lib.NbNoiseSrc(_p(pk["noise_const_t"]), _p(pk["noise_lin"]), _p(layer.noise_strength),
                                   _p(ps.npos if ps.npos_k is None else ps.npos_k), _p(ps.ipos if ps.npos_k is None else None),
                                   s.block_res, self.img_resolution)
            ps.keep_alive.append(nsrc)
            noise_ptr, nstride = ctypes.addressof(nsrc), _lib.NB_NOISE_IN_KERNEL
        elif noise_mode == "const":
            noise_ptr = plan.noise[i].data_ptr()
            nstride = 0 if ps.shared else s.block_res * s.block_res
        elif noise_mode == "random":
            rnd = torch.randn([n, s.block_res, s.block_res], device=device) * layer.noise_strength
            ps.keep_alive.append(rnd)
            noise_ptr, nstride = rnd.data_ptr(), s.block_res * s.block_res
        clamp = -1.0 if layer.conv_clamp is None else float(layer.conv_clamp)
        # split-f16 layers hand activations over in H2 format (pre-multiplied by the consumer's styles).
        # `x_h2` is this layer's complete H2 input if the previous layer produced it; `next_h2` is the
        # consumer's input tensor this layer writes into directly when both ends are split-f16 kernels
        # and nothing taps the fp32 activations in between (feature taps, blending, ToRGB, stop_after).
        nxt = cfg.layers[i + 1] if i + 1 < len(cfg.layers) else None
        at_block_end = s.up == 1
        tapped = at_block_end and (block.is_last or res in return_features or res in blended_features
                                   or stop_after == res)
        me_h3 = self._h3_up2_eligible(s) if s.up == 2 else self._h3_eligible(s)
        nxt_h3 = nxt is not None and (self._h3_eligible(nxt) if nxt.up == 1 else self._h3_up2_eligible(nxt))
        in_fmt = self._operand_fmt(s)                       # 0 = H2 (hi/lo f16), 1 = f8 corrections
        out_fmt = self._operand_fmt(nxt) if nxt_h3 else 0
        geo_after = (self.geom_feature_channels[self.geom_feature_resolutions.index(res)]
                     if at_block_end and res in self.geom_feature_resolutions else 0)
        # (the f6 operand format is written by the f8 / f6 up=1 loops only: a producer on H2 operands -- c_in not a
        #  multiple of 16 -- hands its output over in fp32 and the pack launch writes the consumer's operands)
        fuse_out = (self.h2_handoff and me_h3 and nxt_h3 and not tapped and s.out_channels % 8 == 0
                    and (out_fmt == 0 or (s.out_channels % 16 == 0 and geo_after % 16 == 0))
                    and (out_fmt != 2 or (in_fmt != 0 and s.up == 1)))
        y = next_h2 = None
        ps.fused_rgb = None
        if me_h3:
            if x_h2 is None:
                # producer was not a split-f16 kernel: (x ++ geometry) * styles -> H2 / f8 operands
                evp = self._begin_event("pack_h2")
                x_h2 = torch.empty(ops.h2_shape(n, s.in_channels, s.in_res, s.in_res), dtype=torch.float16,
                                   device=device)
                pack = (lib.nb_pack_h2_f32, lib.nb_pack_h2f8_f32, lib.nb_pack_h2f6_f32)[in_fmt]
                _lib.check(pack(_p(x), c1, _p(x2), c2, _p(plan.styles[i]), _p(x_h2), n, s.in_res * s.in_res, stream),
                           "pack_h2")
                self._end_event(evp)
            ev = self._begin_event(name)
            wts = pk["w_f6"] if in_fmt == 2 else pk["w_f8"] if in_fmt else pk["w_h3"]
            fuse_rgb = (self.fuse_torgb and block.is_last and s.up == 1 and s.out_channels <= 128
                        and res not in blended_features)
            targs = None
            if fuse_rgb:
                # last conv + ToRGB + compositing in one launch; the fp32 activations are only written
                # when a caller taps them
                tg = self._torgb_setup(plan, n, device, extra)
                targs = self._torgb_args(plan, tg, s.out_channels)
                if res in return_features or stop_after == res:
                    y = torch.empty([n, s.out_channels, s.block_res, s.block_res], dtype=torch.float32, device=device)
            elif fuse_out:
                if at_block_end and res in ps.pre_h2:
                    next_h2 = ps.pre_h2[res][0]                    # geometry channels are (being) packed into it
                else:
                    next_h2 = torch.empty(ops.h2_shape(n, nxt.in_channels, s.block_res, s.block_res),
                                          dtype=torch.float16, device=device)
            else:
                y = torch.empty([n, s.out_channels, s.block_res, s.block_res], dtype=torch.float32, device=device)
            nst = _p(plan.styles[i + 1]) if next_h2 is not None else None
            c_next = nxt.in_channels if next_h2 is not None else 0
            kfmt = 3 if (self.conv_mode == "f16" and in_fmt == 1) else in_fmt        # f8 operands, hi x hi products only (large kernels)
            if s.up == 1:
                _lib.check(lib.nb_modconv3x3_up1_h3_ex(
                    _p(x_h2), s.in_channels, _p(wts), _p(plan.dcoefs[i]), noise_ptr, nstride, _p(layer.bias),
                    _p(y), _p(next_h2), nst, c_next, c_next, None if targs is None else ctypes.byref(targs),
                    kfmt, out_fmt if next_h2 is not None else 0, n, s.in_res, s.in_res, s.out_channels, 0.2,
                    layer.act_gain, clamp, stream), name)
            else:
                _lib.check(lib.nb_modconv3x3_up2_h3_ex(
                    _p(x_h2), s.in_channels, _p(wts), _p(plan.dcoefs[i]), noise_ptr, nstride, _p(layer.bias),
                    _p(y), _p(next_h2), nst, c_next, c_next, kfmt, out_fmt if next_h2 is not None else 0, n,
                    s.in_res, s.in_res, s.out_channels, 0.2, layer.act_gain, clamp, stream), name)
            if fuse_rgb:
                ps.fused_rgb = self._torgb_finish(tg, extra)
            self.layer_formats[name] = in_fmt
            self.layer_kernels[name] = ("modconv3x3_up1_h3_kernel<%d>" % (2 if s.out_channels > 64 else 1)
                                        if s.up == 1 else self._up2_h3_variant_name(in_fmt, n, s))
            ps.keep_alive.append(x_h2)
            self._end_event(ev)
        elif self._small_h3_eligible(s) and c2 == 0 and x is not None:
            # small conv1 layer: split-f16 products on 32 x 32 tiles with K split over the waves
            ev = self._begin_event(name)
            y = torch.empty([n, s.out_channels, s.block_res, s.block_res], dtype=torch.float32, device=device)
            _lib.check(lib.nb_modconv3x3_up1_small_h3(
                _p(x), c1, _p(pk["w_h3"]), _p(plan.styles[i]), _p(plan.dcoefs[i]), noise_ptr, nstride,
                _p(layer.bias), _p(y), n, s.in_res, s.in_res, s.out_channels, 0.2, layer.act_gain, clamp, stream), name)
            self.layer_kernels[name] = "modconv3x3_up1_small_h3_kernel"
            self._end_event(ev)
        elif self._small_h3_up2_eligible(s) and c1 % 16 == 0 and c2 % 16 == 0 and x is not None:
            # small conv0 layer: the FIR is folded into four per-phase 3x3 kernels (ops.fold_up2_fir), the
            # phases run through the same small-tile split-f16 kernel
            ev = self._begin_event(name)
            y = torch.empty([n, s.out_channels, s.block_res, s.block_res], dtype=torch.float32, device=device)
            _lib.check(lib.nb_modconv3x3_up2_small_h3(
                _p(x), c1, _p(x2), c2, _p(pk["w_h3_up2"]), _p(plan.styles[i]), _p(plan.dcoefs[i]), noise_ptr, nstride,
                _p(layer.bias), _p(y), n, s.in_res, s.in_res, s.out_channels, 0.2, layer.act_gain, clamp, stream), name)
            self.layer_kernels[name] = "modconv3x3_up1_small_h3_kernel"
            self._end_event(ev)
        else:
            ev = self._begin_event(name)
            y = torch.empty([n, s.out_channels, s.block_res, s.block_res], dtype=torch.float32, device=device)
            _lib.check(lib.nb_modconv3x3_f32(
                _p(x), c1, _p(x2), c2, _p(pk["wpk"]), _p(plan.styles[i]), _p(plan.dcoefs[i]), noise_ptr,
                nstride, _p(layer.bias), _p(y), n, s.in_res, s.in_res, s.out_channels, s.up, 0.2,
                layer.act_gain, clamp, stream), name)
            self.layer_kernels[name] = self._variant_name(n, s)
            self._end_event(ev)
        ps.x_h2 = next_h2
        ps.keep_alive += [x, x2]
        ps.x, ps.x2 = y, None

    def _finish_block(self, ps: "_Pass", block, res: int, specs) -> None:
        """What follows a block's last layer (networks_modified.py:168-222): ToRGB of the last block, feature taps, blending, and the
        geometry feature that is concatenated to the next block's input."""
        opts, plan, n, device, lib, stream = ps.opts, ps.plan, ps.n, ps.device, _lib.lib(), ps.stream
        extra, pre_h2 = opts.extra, ps.pre_h2
        if block.is_last:
            ps.img, triad = ps.fused_rgb if ps.fused_rgb is not None else self._torgb(plan, ps.x, n, stream, extra)
            if ps.return_debug_data:
                ps.debug_data.update(triad)
        if res in ps.return_features:
            ps.debug_data["features%d_preblend" % res] = ps.x
        if res in ps.blended_features:
            bf = ps.blended_features[res]
            ps.x = ops.blend(bf.features.to(device=device, dtype=torch.float32),
                             bf.alpha.to(device=device, dtype=torch.float32), ps.x)
            if block.is_last:                                   # networks_modified.py:182-185
                ps.img, triad = self._torgb(plan, ps.x, n, stream, extra)
                ps.debug_data.update(triad)
        if res in ps.return_features:
            ps.debug_data["features%d" % res] = ps.x
        if res not in self.geom_feature_resolutions:
            return
        x_h2 = ps.x_h2
        g = ps.geom_feature[ps.geo_idx]
        if g is None and x_h2 is not None and res in pre_h2 and x_h2 is pre_h2[res][0]:
            ps.geo_idx += 1                                     # the encoder wrote these channels into x_h2 itself
            return
        if g is None:                                           # (the producer did not take the hand-off after all)
            g = ps.lazy_geom.plain()[ps.geo_idx]
        ps.geo_idx += 1
        if g.device != device:
            raise RuntimeError(f"geom_feature is on {g.device} but the generator is on {device}")
        x2 = g.to(torch.float32).contiguous()
        _assert_shape(x2, [n, self.geom_feature_channels[self.geom_feature_resolutions.index(res)], res, res])
        if x_h2 is not None and res in pre_h2 and x_h2 is pre_h2[res][0]:
            if pre_h2[res][1] is not None and not ps.packs_waited:
                # packed early on the side stream.  ONE wait, at the first consumer, for the LAST pack enqueued there (the
                # stream runs them in order, and all of them are through long before this point: 131 us into a step whose
                # first consumer starts at 168): a cross-stream wait costs ~6 us of idle chip each time, whether or not
                # the event has fired (tools/trace_step_timeline.py)
                last_ = getattr(plan, "pack_last_event", None)
                covers_all = any(v[1] is last_ for v in pre_h2.values())        # (the last pack of THIS pass)
                torch.cuda.current_stream(device).wait_event(last_ if covers_all else pre_h2[res][1])
                ps.packs_waited = covers_all
            ps.keep_alive.append(x2)
            x2 = None
        elif x_h2 is not None:
            # the block's last layer already wrote its channels into the consumer's H2 input: add the
            # geometry channels (x the consumer's styles) behind them
            inext, snext = specs[f"synthesis.b{2 * res}.conv0"]
            c_prod = snext.in_channels - x2.shape[1]
            evp = self._begin_event("pack_h2")
            part = (lib.nb_pack_h2_part_f32, lib.nb_pack_h2f8_part_f32, lib.nb_pack_h2f6_part_f32)[self._operand_fmt(snext)]
            _lib.check(part(_p(x2), x2.shape[1], plan.styles[inext].data_ptr() + 4 * c_prod, snext.in_channels,
                            _p(x_h2), (snext.in_channels + 7) // 8, c_prod // 8, n, res * res, stream),
                       "pack_h2_part")
            self._end_event(evp)
            ps.keep_alive.append(x2)
            x2 = None
        ps.x2 = x2

    def _torgb_setup(self, plan: _Plan, n, dev, extra):
        """Allocate the ToRGB outputs and collect the launch arguments (shared by the standalone and fused forms)."""
        cfg = self.cfg
        t = self.get_last_block().torgb
        r = cfg.img_resolution
        uvs = torch.empty([n, 3, r, r], dtype=torch.float32, device=dev)
        img = torch.empty([n, 3, r, r], dtype=torch.float32, device=dev)
        colors = torch.empty([n, 3, 3], dtype=torch.float32, device=dev)
        logits = rgba = rgba8 = user = sfac = None
        mode = 0
        if extra is not None:
            if extra.get("logits"):
                logits = torch.empty([n, 3, r, r], dtype=torch.float32, device=dev)
            if extra.get("rgba"):
                rgba = torch.empty([n, 4, r, r], dtype=torch.float32, device=dev)
            if extra.get("rgba_u8"):
                rgba8 = torch.empty([n, r, r, 4], dtype=torch.uint8, device=dev)
            user = extra.get("user_colors")
            if user is not None:
                user = user.to(device=dev, dtype=torch.float32).contiguous()
                _assert_shape(user, [n, 3, 3])
            sfac = extra.get("sfactor")
            if sfac is not None:
                sfac = torch.as_tensor(sfac, dtype=torch.float32, device=dev).reshape(-1)
                sfac = (sfac.expand(n) if sfac.numel() == 1 else sfac).contiguous()
                _assert_shape(sfac, [n])
            mode = {"clear": 0, "full": 1}.get(extra.get("render_mode", "clear"), -1)
            if mode < 0:
                raise RuntimeError("Unknown render mode for TriadGanPaintEngine: {}".format(extra.get("render_mode")))
        clamp = -1.0 if t.conv_clamp is None else float(t.conv_clamp)
        return dict(t=t, uvs=uvs, img=img, colors=colors, logits=logits, rgba=rgba, rgba8=rgba8, user=user, sfac=sfac,
                    mode=mode, clamp=clamp, r=r)

    def _torgb_finish(self, o, extra):
        if extra is not None:
            extra["out"] = {"logits": o["logits"], "rgba": o["rgba"], "rgba_u8": o["rgba8"]}
        return o["img"], {"colors": o["colors"], "uvs": o["uvs"]}

    def _torgb_args(self, plan: _Plan, o, c) -> "_lib.NbTorgbArgs":
        a = _lib.NbTorgbArgs()
        a.styles, a.w, a.bias, a.color_bias = _p(plan.styles[-1]), _p(self.packed["torgb"]["w"]), _p(o["t"].bias), _p(o["t"].color_bias)
        a.logits, a.uvs, a.img, a.colors_out = _p(o["logits"]), _p(o["uvs"]), _p(o["img"]), _p(o["colors"])
        a.user_colors, a.sfactor, a.rgba_f32, a.rgba_u8 = _p(o["user"]), _p(o["sfac"]), _p(o["rgba"]), _p(o["rgba8"])
        a.styles_stride_n, a.render_mode, a.clamp = c + 9, o["mode"], o["clamp"]
        return a

    def _torgb(self, plan: _Plan, x, n, stream, extra):
        c = x.shape[1]
        o = self._torgb_setup(plan, n, x.device, extra)
        r = o["r"]
        ev = self._begin_event("torgb")
        _lib.check(_lib.lib().nb_torgb_triad_f32(
            _p(x), _p(plan.styles[-1]), c + 9, _p(self.packed["torgb"]["w"]), _p(o["t"].bias), _p(o["t"].color_bias),
            o["clamp"], _p(o["logits"]), _p(o["uvs"]), _p(o["img"]), _p(o["colors"]), _p(o["user"]), _p(o["sfac"]),
            o["mode"], _p(o["rgba"]), _p(o["rgba8"]), n, c, r * r, stream), "torgb_triad")
        self._end_event(ev)
        return self._torgb_finish(o, extra)


class _SplitForward:
    """Results of a forward that runs as sub-batches on side streams (``Generator._forward_split``).  ``join()`` makes
    the caller's stream wait for them and returns what the unsplit call returns; until then the parts are only valid
    on their own streams (a throughput loop can keep enqueueing steps and synchronise the device once at the end)."""

    def __init__(self, G, main, parts, extras, extra, ws):
        self.G, self.main, self.parts, self.extras, self.extra, self.ws = G, main, parts, extras, extra, ws

    def join(self):
        main = self.main
        for st in self.G._side_streams:
            main.wait_stream(st)

        def cat(ts):
            if ts[0] is None:
                return None
            with torch.cuda.stream(main):
                out = torch.cat(ts)
            for t in ts:
                t.record_stream(main)                 # produced on a side stream, read on the main one
            return out
        if self.extra is not None:
            self.extra["out"] = {k: cat([e["out"][k] for e in self.extras]) for k in self.extras[0]["out"]}
        parts = self.parts
        if isinstance(parts[0], tuple):
            img = cat([p[0] for p in parts])
            dbg = {k: cat([p[1][k] for p in parts]) for k in parts[0][1]}
            if self.ws is not None:
                dbg["ws"] = self.ws
            return img, dbg
        return cat(list(parts))


def mix_styles(mapping, ws, z, c, style_mixing_prob, **mapping_kwargs):
    """Style mixing regularisation (networks_modified.py:385-394): with probability ``style_mixing_prob`` the W+ rows from a random
    cutoff onwards come from a second latent.  Shared by the inference ``Generator`` and ``training.TrainableGenerator``."""
    cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
    cutoff = torch.where(torch.rand([], device=ws.device) < style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
    try:
        ws2 = mapping(torch.randn_like(z), c, skip_w_avg_update=True, **mapping_kwargs)
    except TypeError:                                  # (the inference mapping network has no moving average to skip)
        ws2 = mapping(torch.randn_like(z), c, **mapping_kwargs)
    sel = (torch.arange(ws.shape[1], device=ws.device) >= cutoff)[None, :, None]
    return torch.where(sel, ws2.to(ws.dtype), ws)


class Generator(torch.nn.Module):
    """``networks_modified.py:227-400``."""

    def __init__(self, cfg: GeneratorConfig = None, state_dict: Optional[StateDict] = None,
                 conv_mode: str = DEFAULT_CONV_MODE, **kwargs):
        super().__init__()
        if conv_mode not in CONV_MODES:
            raise RuntimeError(f"unknown conv_mode {conv_mode!r}")
        if cfg is None:
            cfg = GeneratorConfig(**kwargs)
        self.cfg = cfg
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, cfg.c_dim, cfg.w_dim
        self.img_resolution, self.img_channels = cfg.img_resolution, cfg.img_channels
        self.positional_encoder = None
        self.synthesis = SynthesisNetwork(cfg)
        self.synthesis.conv_mode = conv_mode
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(cfg)
        self.geom_inject = True
        self._side_streams = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())
        if state_dict is not None:
            self.load_numpy_state_dict(state_dict)
        self.eval().requires_grad_(False)

    def set_conv_mode(self, conv_mode: str):
        """'h3': layers >= 32x32 as 3-pass split-f16 MFMA (5e-6 from fp32); 'f8': the two correction passes on one
        block-scaled fp8 MFMA per tap pair (1e-4 from fp32, ~1.4x faster layers); 'f32': exact-fp32 MFMA kernels."""
        if conv_mode not in CONV_MODES:
            raise RuntimeError(f"unknown conv_mode {conv_mode!r}")
        self.synthesis.conv_mode = conv_mode
        self._invalidate()
        return self

    def _invalidate(self):
        self.synthesis.invalidate()
        self.mapping._packed = None

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_numpy_state_dict(self, sd: StateDict):
        validate_state_dict(self.cfg, sd)
        self.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32))) for k, v in sd.items()},
                             strict=True)
        return self

    @staticmethod
    def from_random(cfg: GeneratorConfig, seed: int = 0, device="cuda") -> "Generator":
        return Generator(cfg, random_state_dict(cfg, seed)).to(device)

    def forward_pre_mapped(self, ws, geom_feature, positions=None, return_debug_data=False, return_features=None,
                           blended_features=None, noise_buffers=None, **synthesis_kwargs):
        # networks_modified.py:351-353 normalises positions here; this build hands the integer positions to
        # nb_noise_f32, which does the same (positions % R)/(R-1) in correctly rounded fp32 (see neube_hip.h)
        n = ws.shape[0]
        if (self.sub_streams > 1 and n >= self.sub_stream_min_batch and not blended_features and not noise_buffers
                and "_plan_slot" not in synthesis_kwargs and ws.is_cuda and not torch.cuda.is_current_stream_capturing()):
            return self._forward_split(ws, geom_feature, positions, return_debug_data, return_features, synthesis_kwargs)
        synthesis_kwargs.pop("_join", None)
        syn_res = self.synthesis(ws, geom_feature, pos_encoding=None, return_debug_data=return_debug_data,
                                 return_features=return_features, blended_features=blended_features,
                                 **synthesis_kwargs, _positions=positions, noise_buffers=noise_buffers)
        if return_debug_data or return_features:
            img, debug_data = syn_res
            if return_debug_data:
                debug_data["ws"] = ws
            return img, debug_data
        return syn_res

    # Two sub-batches in flight on two HIP streams: every launch of a layer ends with a partially filled last round
    # of workgroups and the next layer cannot start before it drains; with a second, independent chain of launches
    # the CUs that fall idle at one chain's kernel boundary pick up the other chain's workgroups.  Worth +5 % at batch 64;
    # at batch 32 it was +7.6 % in round 1 and is +0.5 % since the small-image layers moved to the large kernels and the
    # epilogues were rewritten (tools/sub_stream_sweep.py), so a batch of 32 runs as ONE chain: every launch then has the
    # chip to itself and its HIP-event duration is the kernel's own time.
    # (Below 256x256 the launches are a quarter of the size and the second chain still pays: R=128, batch 32: 34 900 vs
    # 31 400 patches/s.)
    sub_streams = 2
    _sub_stream_min_batch = None             # set to override the rule below

    @property
    def sub_stream_min_batch(self) -> int:
        if self._sub_stream_min_batch is not None:
            return self._sub_stream_min_batch
        return 64 if self.img_resolution >= 256 else 16

    @sub_stream_min_batch.setter
    def sub_stream_min_batch(self, v: int):
        self._sub_stream_min_batch = v

    def _forward_split(self, ws, geom_feature, positions, return_debug_data, return_features, kw):
        n = ws.shape[0]
        dev = ws.device
        lazy_geom = geom_feature if hasattr(geom_feature, "encode_for") else None      # encoder.LazyGeometry: split per sub-batch
        if lazy_geom is not None:
            geom_feature = []
        else:
            geom_feature = list(geom_feature) if isinstance(geom_feature, (list, tuple)) else [geom_feature]
        kw = dict(kw)
        extra = kw.pop("_extra_outputs", None)
        resume = kw.pop("_resume", None)
        npos = kw.pop("norm_noise_positions", None)
        kw_join = kw.pop("_join", True)
        if self._side_streams is None or self._side_streams[0].device != dev:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(self.sub_streams)]
        main = torch.cuda.current_stream(dev)
        bounds = [(i * n // self.sub_streams, (i + 1) * n // self.sub_streams) for i in range(self.sub_streams)]
        sl = lambda t, a, b: None if t is None else (t[a:b] if (torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == n) else t)
        parts, extras = [], []
        for i, (a, b) in enumerate(bounds):
            st = self._side_streams[i]
            st.wait_stream(main)
            ex = None
            if extra is not None:
                ex = {k: (sl(torch.as_tensor(v), a, b) if k in ("user_colors", "sfactor") and v is not None else v)
                      for k, v in extra.items() if k != "out"}
            for t in [ws, positions, npos, None if resume is None else resume[1]] + geom_feature:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(st)               # allocated on the caller's stream, read on this one
            if lazy_geom is not None and torch.is_tensor(lazy_geom.geom) and lazy_geom.geom.is_cuda:
                lazy_geom.geom.record_stream(st)
            with torch.cuda.stream(st):
                res = self.synthesis(ws[a:b], lazy_geom.sliced(a, b) if lazy_geom is not None else [g[a:b] for g in geom_feature],
                                     pos_encoding=None,
                                     return_debug_data=return_debug_data, return_features=return_features, **kw,
                                     _positions=sl(positions, a, b), norm_noise_positions=sl(npos, a, b),
                                     _resume=None if resume is None else (resume[0], resume[1][a:b]),
                                     _extra_outputs=ex, _plan_slot=i + 1)
            parts.append(res)
            extras.append(ex)
        pending = _SplitForward(self, main, parts, extras, extra, ws if return_debug_data else None)
        if kw_join is False:
            return pending                            # caller joins (or synchronises the device) before using the outputs
        return pending.join()

    def forward(self, z, c, geom_feature, positions=None, noise_buffers=None, truncation_psi=1, truncation_cutoff=None,
                return_debug_data=False, return_features=None, blended_features=None, style_mixing_prob=0,
                **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        if style_mixing_prob > 0:
            ws = mix_styles(self.mapping, ws, z, c, style_mixing_prob, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.forward_pre_mapped(ws, geom_feature, positions=positions, return_debug_data=return_debug_data,
                                       return_features=return_features, blended_features=blended_features,
                                       noise_buffers=noise_buffers, **synthesis_kwargs)

    def render_triad(self, z=None, ws=None, geom_feature=None, positions=None, render_mode="clear", user_colors=None,
                     want_u8=True, want_f32=False, sfactor=None, join=True, **kw):
        """Generator + the paint engine's compositing (brush.py:763-792) fused into the ToRGB launch.
        Returns (rgba_u8 [N,R,R,4] | None, rgba_f32 [N,4,R,R] | None, debug dict with uvs/colors).
        ``join=False`` (throughput loops): when the batch runs as sub-batches on side streams, returns a callable that
        joins and yields that tuple; steps enqueued without joining overlap across the streams."""
        extra = {"rgba_u8": want_u8, "rgba": want_f32, "render_mode": render_mode, "user_colors": user_colors,
                 "sfactor": sfactor}
        kw.setdefault("noise_mode", "const")
        if ws is None:
            res = self.forward(z, None, geom_feature, positions=positions, return_debug_data=True,
                               _extra_outputs=extra, _join=join, **kw)
        else:
            res = self.forward_pre_mapped(ws, geom_feature, positions=positions, return_debug_data=True,
                                          _extra_outputs=extra, _join=join, **kw)
        if isinstance(res, _SplitForward):
            def finish():
                _, dbg = res.join()
                return extra["out"]["rgba_u8"], extra["out"]["rgba"], dbg
            # per sub-batch results, each valid on its own stream (e.g. to start a gather of a part from that stream)
            finish.streams = list(self._side_streams)
            finish.parts_u8 = [e["out"]["rgba_u8"] for e in res.extras]
            return finish
        img, dbg = res
        return extra["out"]["rgba_u8"], extra["out"]["rgba"], dbg
