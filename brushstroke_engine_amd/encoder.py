"""Geometry encoder of the painting engine.

The reference's ``sauto`` autoencoder (``forger/experimental/autoenc/simple_autoencoder.py:155-199, 251-261,
289-297``; ``base.py:123-134``) turns the stroke-geometry patch [N,1,R,R] (1 = background, 0 = stroke) into the two
feature maps the generator consumes: the 16-channel bottleneck at R/8 and the first decoder stage (bilinear x2 +
conv, 256 channels) at R/4.  It is a plain conv / eval-mode BatchNorm / LeakyReLU(0.01) stack with reflect padding.

Evaluated by hand-written gfx950 kernels only (``HipGeometryEncoder`` on ``csrc/nb_encoder.hip``): there is no
PyTorch / MIOpen module of it in this package (patch sizes the kernels do not tile raise).  State-dict keys equal the
reference's (``encoder.model.{i}.conv.{0,1}.*``, ``decoder.model.{i}...``) so ``strong.pt``-style checkpoints load unchanged.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os
import numpy as np
import torch


# state_dict keys and shapes of the reference's ``sauto`` autoencoder (simple_autoencoder.py:155-199, defaults of
# add_model_flags: pre 64 @7x7, down 128,256,256, post 32,16, up 256,128,64, 1 channel in / out); every _SingleConvolution is
# `conv.0` (Conv2d) + `conv.1` (BatchNorm2d), decoder stages are ScaleUp(`conv` = _SingleConvolution), `decoder.model.3` is the
# final 1x1 conv.  Checked against the instantiated reference in tests/golden/make_golden_engine.py (strict load).
def _conv_bn(prefix, o, i, k):
    return [(f"{prefix}.0.weight", (o, i, k, k)), (f"{prefix}.0.bias", (o,)), (f"{prefix}.1.weight", (o,)), (f"{prefix}.1.bias", (o,)),
            (f"{prefix}.1.running_mean", (o,)), (f"{prefix}.1.running_var", (o,)), (f"{prefix}.1.num_batches_tracked", ())]


ENCODER_STATE_SHAPES = (
    _conv_bn("encoder.model.0.conv", 64, 1, 7) + _conv_bn("encoder.model.1.conv", 128, 64, 3) + _conv_bn("encoder.model.2.conv", 256, 128, 3)
    + _conv_bn("encoder.model.3.conv", 256, 256, 3) + _conv_bn("encoder.model.4.conv", 32, 256, 3) + _conv_bn("encoder.model.5.conv", 16, 32, 3)
    + _conv_bn("decoder.model.0.conv.conv", 256, 16, 3) + _conv_bn("decoder.model.1.conv.conv", 128, 256, 3)
    + _conv_bn("decoder.model.2.conv.conv", 64, 128, 3) + [("decoder.model.3.weight", (1, 64, 1, 1)), ("decoder.model.3.bias", (1,))])


def random_encoder_state_dict(seed: int = 5) -> Dict[str, np.ndarray]:
    """Seeded synthetic encoder weights with the reference's key names (no encoder checkpoint ships with it)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, shp in ENCODER_STATE_SHAPES:
        if k.endswith("num_batches_tracked"):
            sd[k] = np.zeros(shp, np.int64)
        elif k.endswith("running_var"):
            sd[k] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif k.endswith("running_mean") or k.endswith(".bias"):
            sd[k] = (0.1 * rs.randn(*shp)).astype(np.float32)
        elif len(shp) == 4:                       # conv weight, He-ish scale keeps activations O(1)
            fan_in = shp[1] * shp[2] * shp[3]
            sd[k] = (rs.randn(*shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        else:                                     # BatchNorm weight
            sd[k] = rs.uniform(0.8, 1.2, shp).astype(np.float32)
    return sd


# ------------------------------------------------------------------------------------------------
# hand-written HIP path (csrc/nb_encoder.hip)
# ------------------------------------------------------------------------------------------------
def _fold_bn(sd, prefix: str):
    """conv + eval BatchNorm -> conv' : w' = w*s, b' = (b - mean)*s + beta, s = gamma / sqrt(var + 1e-5)."""
    g = lambda k: np.asarray(sd[prefix + k], np.float64)
    s = g(".conv.1.weight") / np.sqrt(g(".conv.1.running_var") + 1e-5)
    w = g(".conv.0.weight") * s[:, None, None, None]
    b = (g(".conv.0.bias") - g(".conv.1.running_mean")) * s + g(".conv.1.bias")
    return w.astype(np.float32), b.astype(np.float32)


def pack_enc_weight_h3(w: np.ndarray) -> np.ndarray:
    """[O,I,3,3] fp32 -> hi/lo f16 [ceil(I/16)][ky][kx][cg 2][hi/lo 2][ceil128(O)][8] (include/neube_hip.h)."""
    o, i = w.shape[:2]
    nch, co_ld = -(-i // 16), -(-o // 128) * 128
    wp = np.zeros([nch * 16, 3, 3, co_ld], np.float32)
    wp[:i, :, :, :o] = w.transpose(1, 2, 3, 0)
    hi = wp.astype(np.float16)
    lo = (wp - hi.astype(np.float32)).astype(np.float16)
    a = np.stack([hi, lo], axis=0).reshape(2, nch, 2, 8, 3, 3, co_ld)       # [hl][chunk][cg][8][ky][kx][co]
    return np.ascontiguousarray(a.transpose(1, 4, 5, 2, 0, 6, 3))           # [chunk][ky][kx][cg][hl][co][8]


def pack_enc_weight_f8(w: np.ndarray) -> np.ndarray:
    """[O,I,3,3] fp32 (I % 16 == 0) -> the "f8" weight operands in the same containers as :func:`pack_enc_weight_h3`: hi slots
    = f16(w) of 8 channels; lo slot of chunk group 0 = fp8 e4m3(w), of group 1 = fp8((w - f16(w)) * 2^11), each over the 16
    channels of the chunk (the weight side of nb_modconv_h3.hip's f8 arithmetic)."""
    o, i = w.shape[:2]
    assert i % 16 == 0
    nch, co_ld = i // 16, -(-o // 128) * 128
    wp = np.zeros([i, 3, 3, co_ld], np.float32)
    wp[:, :, :, :o] = w.transpose(1, 2, 3, 0)
    hi = wp.astype(np.float16)
    wl = (wp - hi.astype(np.float32)) * 2048.0
    f8 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    # [chunk][16 ch][ky][kx][co] -> per (chunk, ky, kx, co): 16 bytes
    b_w = f8(wp).reshape(nch, 16, 3, 3, co_ld).transpose(0, 2, 3, 4, 1)           # [chunk][ky][kx][co][16]
    b_wl = f8(wl).reshape(nch, 16, 3, 3, co_ld).transpose(0, 2, 3, 4, 1)
    out = np.zeros([nch, 3, 3, 2, 2, co_ld, 8], np.float16)
    out[:, :, :, :, 0] = hi.reshape(nch, 2, 8, 3, 3, co_ld).transpose(0, 3, 4, 1, 5, 2)   # [chunk][ky][kx][cg][co][8]
    ob = out.view(np.uint8).reshape(nch, 3, 3, 2, 2, co_ld, 16)
    ob[:, :, :, 0, 1] = b_w
    ob[:, :, :, 1, 1] = b_wl
    return out


class HipGeometryEncoder:
    """``AutoEncoder.encode(geom, res=[0, 1])`` on the hand-written gfx950 kernels: 8 launches per batch
    (stem, 3 stride-2 convs, 2 bottleneck convs, bilinear x2, decoder conv; 7 for f8 batches, whose stem is computed inside
    the first stride-2 launch), BatchNorm folded on the host.
    Interface of the reference's ``AutoEncoder`` as the engine uses it (``encode``, ``feature_channels``,
    ``featuremap_resolution``).  Patch sizes: 32, 64 and multiples of 128 (``supports``)."""

    _PRE = {None: 0, "none": 0, "-11inverse": 1, "inverse": 2}
    arith = "f8"           # operand format between the layers for large batches ("h3": hi/lo f16 everywhere)
    f8_min_batch = 8       # below this the launches are under-filled and the H2-reading small-tile kernel takes over
    fuse_stem = os.environ.get("NB_ENC_FUSE_STEM", "1") != "0"      # f8 batches: stem + first stride-2 stage in one launch

    def __init__(self, state_dict: Dict[str, np.ndarray], preproc_type=None, device="cuda"):
        from . import _lib
        if preproc_type not in self._PRE:
            raise RuntimeError(f'Unknown preprocessing type "{preproc_type}"')
        self._lib = _lib
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NeubeHipError("HipGeometryEncoder needs a GPU (no CPU path in this build)")
        self.preproc_type = preproc_type
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        w, b = _fold_bn(state_dict, "encoder.model.0")
        w50 = np.zeros([64, 50], np.float32)
        w50[:, :49] = w.reshape(64, 49)
        self.stem = (dev(w50), dev(b))
        self.convs = []                                   # (packed weight, bias, c_in, c_out, stride)
        for prefix, stride in (("encoder.model.1", 2), ("encoder.model.2", 2), ("encoder.model.3", 2),
                               ("encoder.model.4", 1), ("encoder.model.5", 1), ("decoder.model.0.conv", 1)):
            w, b = _fold_bn(state_dict, prefix)
            self.convs.append((dev(pack_enc_weight_h3(w)).view(torch.float16), dev(b), w.shape[1], w.shape[0], stride,
                               dev(pack_enc_weight_f8(w)).view(torch.float16)))

    def feature_channels(self, res=0):
        return [16, 256, 128, 64][res]

    def featuremap_resolution(self, input_res, res=0):
        return (input_res // 8) * (2 ** res)

    @staticmethod
    def supports(resolution: int) -> bool:
        """Patch sizes the kernels tile: multiples of 128 (every layer on the large tiles) and the powers of two below -- 32
        and 64, the remaining sizes a StyleGAN2 generator can have -- whose inner layers (outputs 4 or 8 pixels wide) take the
        32-position split-K tiles (``enc_conv3x3_small_h3_kernel``), which read hi/lo-f16 operands."""
        return resolution in (32, 64) or (resolution >= 128 and resolution % 128 == 0)

    @staticmethod
    def large_tiles_only(resolution: int) -> bool:
        return resolution >= 128 and resolution % 128 == 0

    def can_handoff(self, resolution: int, i: int) -> bool:
        """Can feature i of a patch of this size be written straight into a generator layer's operand tensor (large tiles)?"""
        r = self.featuremap_resolution(resolution, i)
        return i == 1 and (r % 32 == 0 or r == 16)

    def lazy(self, geom: torch.Tensor) -> "LazyGeometry":
        """The geometry features of ``geom`` as a provider the HIP generator evaluates itself, right after it has computed
        its styles: the 256-channel decoder feature then goes straight into the consuming layer's operand tensor
        (``nb_enc_conv3x3_h3_handoff``) instead of fp32 NCHW + a packing pass."""
        return LazyGeometry(self, geom)

    @torch.no_grad()
    def encode(self, geom: torch.Tensor, res=None, targets=None) -> List[torch.Tensor]:
        """``targets`` (optional, from the generator): ``{1: dict(dst, scale_ptr, scale_stride, c8_total, cg0, fmt)}`` --
        feature 1 is then written into ``dst`` (the consumer's H2 / f8 input tensor) and returned as None."""
        if res is not None and list(res) != [0, 1]:
            raise RuntimeError("HipGeometryEncoder evaluates the shipped configuration: res=[0, 1]")
        lib, check = self._lib.lib(), self._lib.check
        n, c, h, w = geom.shape
        if c != 1 or not self.supports(h) or h != w:
            raise RuntimeError(f"HipGeometryEncoder: unsupported geometry shape {tuple(geom.shape)} (square; R = 32, 64 or a multiple of 128)")
        x = geom.to(self.device, torch.float32).contiguous()
        f16 = lambda ch, r: torch.empty([n, ch // 8, 2, r, r, 8], dtype=torch.float16, device=self.device)
        f32 = lambda ch, r: torch.empty([n, ch, r, r], dtype=torch.float32, device=self.device)
        P = lambda t: t.data_ptr()
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            # operand format between the layers: "f8" (fp8 correction operands, 2/3 of the matrix cycles) for batches that
            # fill the chip with the large-tile kernel; H2 (hi/lo f16) otherwise -- the small-tile kernel of interactive
            # strokes reads H2
            fmt = 1 if (self.arith == "f8" and n >= self.f8_min_batch and self.large_tiles_only(h)) else 0
            W = lambda cv, f=None: P(cv[5] if (fmt if f is None else f) else cv[0])
            # layers with <= 32 output channels (256 -> 32, 32 -> 16) leave three quarters of the large tile's 128 c_out rows empty:
            # the library runs them on the 32 x 32 split-K tiles, which read hi/lo-f16 operands -- so their producers write H2
            narrow = lambda i: self.convs[i][3] <= 32 and self.convs[i][2] >= 32
            # stem + first stride-2 stage: ONE launch for f8 batches (the 64-channel full-resolution tensor between them -- the largest
            # of the pass, written once and read once -- never exists); else the stem kernel and the general conv kernel
            fused = bool(fmt and self.fuse_stem and h % 64 == 0)
            if fused:
                _, b, ci, co, stride, wf8 = self.convs[0]
                a = f16(co, h // 2)
                check(lib.nb_enc_stem_conv3x3_f8(P(x), P(self.stem[0]), P(self.stem[1]), self._PRE[self.preproc_type], P(wf8), P(b), P(a),
                                                 0 if narrow(1) else fmt, n, h, w, co, 0.01, st), "enc_stem_conv")
                r = h // 2
            else:
                a = f16(64, h)
                check(lib.nb_enc_stem7x7_f32_h2_ex(P(x), P(self.stem[0]), P(self.stem[1]), P(a), fmt, n, h, w,
                                                   self._PRE[self.preproc_type], 0.01, st), "enc_stem")
                r = h
            for i in range(1 if fused else 0, 4):            # three stride-2 stages + 256 -> 32
                _, b, ci, co, stride, _ = self.convs[i]
                r_out = r // stride
                y = f16(co, r_out)
                fin = 0 if narrow(i) else fmt
                fout = 0 if narrow(i + 1) else fmt
                check(lib.nb_enc_conv3x3_ex(P(a), ci, W(self.convs[i], fin), P(b), None, P(y), None, 0, co // 8 if fout else 0, 0, fin, fout,
                                            n, r, r, co, stride, 0.01, st), "enc_conv")
                a, r = y, r_out
            _, b, ci, co, stride, _ = self.convs[4]          # 32 -> 16: the bottleneck the generator consumes
            enc = f32(co, r)
            fin = 0 if narrow(4) else fmt
            check(lib.nb_enc_conv3x3_ex(P(a), ci, W(self.convs[4], fin), P(b), P(enc), None, None, 0, 0, 0, fin, 0, n, r, r, co, 1, 0.01, st),
                  "enc_conv")
            up = f16(co, 2 * r)
            check(lib.nb_enc_upsample2x_h2_ex(P(enc), P(up), fmt, n, co, r, r, st), "enc_upsample")
            _, b, ci, co, stride, _ = self.convs[5]          # first decoder stage, 16 -> 256
            tg = None if not targets else targets.get(1)
            if tg is not None and not self.can_handoff(h, 1):
                raise RuntimeError(f"HipGeometryEncoder: no hand-off into operand tensors at patch size {h} (LazyGeometry.can_handoff)")
            if tg is not None:
                dec = None
                check(lib.nb_enc_conv3x3_ex(P(up), ci, W(self.convs[5]), P(b), None, P(tg["dst"]), tg["scale_ptr"], tg["scale_stride"],
                                            tg["c8_total"], tg["cg0"], fmt, tg["fmt"], n, 2 * r, 2 * r, co, 1, 0.01, st), "enc_conv")
            else:
                dec = f32(co, 2 * r)
                check(lib.nb_enc_conv3x3_ex(P(up), ci, W(self.convs[5]), P(b), P(dec), None, None, 0, 0, 0, fmt, 0, n, 2 * r, 2 * r, co, 1,
                                            0.01, st), "enc_conv")
        return [enc, dec]


class LazyGeometry:
    """Geometry features that have not been computed yet (``HipGeometryEncoder.lazy``).  ``networks.SynthesisNetwork``
    calls :meth:`encode_for` once its styles are on the device; everything else that indexes it like the list of feature
    tensors the reference passes (``geom_feature[i]``, ``len``) gets the plain fp32 features, computed on first use."""

    def __init__(self, encoder: "HipGeometryEncoder", geom: torch.Tensor):
        self.encoder, self.geom = encoder, geom
        self._plain = None

    def plain(self) -> List[torch.Tensor]:
        if self._plain is None:
            self._plain = self.encoder.encode(self.geom)
        return self._plain

    def encode_for(self, targets) -> List[Optional[torch.Tensor]]:
        if self._plain is not None or not targets:
            return self.plain()
        return self.encoder.encode(self.geom, targets=targets)

    def sliced(self, a: int, b: int) -> "LazyGeometry":
        """The provider of samples a..b-1 (the generator runs a large batch as sub-batches on separate streams)."""
        return LazyGeometry(self.encoder, self.geom[a:b])

    def can_handoff(self, i: int) -> bool:
        return self.encoder.can_handoff(self.geom.shape[2], i)

    def feature_shape(self, i: int):
        n, _, h, _ = self.geom.shape
        r = self.encoder.featuremap_resolution(h, i)
        return (n, self.encoder.feature_channels(i), r, r)

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return self.plain()[i]

    def __iter__(self):
        return iter(self.plain())
