"""Wire and on-disk formats around the generator: the web client's binary patch protocol, brush
libraries, and the engine snapshot container.

Reference: ``forger/ui/util.py:20-104`` (binary messages of the drawing websocket; must stay compatible with the JS
client's ``encodeDrawingRequest`` / ``decodeDrawingResponse``), ``forger/ui/library.py:49-230`` (brush libraries: seed
lists, W+ pickles with per-layer noise), ``forger/ui/brush.py:552-604`` + ``SG/legacy.py`` (snapshot ``.pkl``).

Snapshots: reference pickles embed Python source of the network classes (``torch_utils/persistence.py``) and can only be
opened with the reference tree importable; ``tools/convert_snapshot.py`` does that once and writes the flat ``.npz``
container of ``weights.save_weights`` (generator tensors under their reference names + ``encoder/<key>`` tensors + JSON
metadata), which is all this build needs at run time.
"""
from __future__ import annotations

import os
import pickle
import random
import re
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# websocket binary protocol (forger/ui/util.py:20-104)
# ------------------------------------------------------------------------------------------------
def int32_to_binary(single_int: int) -> bytes:
    return np.array([single_int], dtype=np.int32).tobytes()


def image_patch_to_binary(img: np.ndarray, x: int, y: int) -> bytes:
    """Response message: int32 [width, height, x, y] followed by the H x W x C uint8 pixels."""
    if img.dtype != np.uint8:
        raise RuntimeError("Image must be uint8 in range 0...255")
    height, width, nchannels = img.shape
    assert nchannels < height, f"Wrong shape {img.shape}"
    return np.array([width, height, x, y], dtype=np.int32).tobytes() + np.ascontiguousarray(img).tobytes()


def binary_to_image_patches(bytes_msg: bytes, offset: int = 0):
    """Request message: int32 [width, height, x, y, crop_margin], then the RGBA stroke patch (the canvas patch that may
    follow is not used).  Returns (meta, stroke [H,W,4] uint8, None)."""
    metadata = np.frombuffer(bytes_msg, dtype=np.int32, count=5, offset=offset)
    meta = {"width": int(metadata[0]), "height": int(metadata[1]), "x": int(metadata[2]), "y": int(metadata[3]),
            "crop_margin": int(metadata[4])}
    img_data = np.frombuffer(bytes_msg, dtype=np.uint8, offset=offset + 5 * 4)
    imgsize = meta["height"] * meta["width"] * 4
    if img_data.size < imgsize:
        raise ValueError(f"message holds {img_data.size} image bytes, header announces {imgsize}")
    return meta, img_data[0:imgsize].reshape((meta["height"], meta["width"], 4)), None


def decode_render_request_metadata(bytes_msg: bytes, offset: int = 0):
    """uint8 [debug, n_colors, extra] then n_colors x [color index, R, G, B].  Returns (meta, next read offset)."""
    metadata = np.frombuffer(bytes_msg, dtype=np.uint8, count=3, offset=offset)
    read_start = offset + 3
    meta = {"debug": bool(metadata[0] != 0), "colors": [], "extra_data": int(metadata[2])}
    for _ in range(int(metadata[1])):
        meta["colors"].append(np.frombuffer(bytes_msg, dtype=np.uint8, count=4, offset=read_start))
        read_start += 4
    return meta, read_start


# ------------------------------------------------------------------------------------------------
# brush libraries (forger/ui/library.py)
# ------------------------------------------------------------------------------------------------
def read_zs(saved_file: str) -> Tuple[List[int], int]:
    """Seed-library file (``forger/ui/library.py`` format): one style per line, the first whitespace-separated token is the
    integer seed, any further tokens are the stored latent; ``#`` starts a comment line.  Returns (seeds in file order,
    number of latent values on the last parsed line).  A missing file is an empty library; a line whose first token is not
    an integer is skipped (the reference logs it and goes on)."""
    if not os.path.isfile(saved_file):
        return [], 0
    records = []
    with open(saved_file) as f:
        for tokens in (ln.split() for ln in f if ln.strip() and not ln.lstrip().startswith("#")):
            if re.fullmatch(r"[+-]?\d+", tokens[0]):
                records.append((int(tokens[0]), len(tokens) - 1))
    return [seed for seed, _ in records], (records[-1][1] if records else 0)


def _interp_style_id(style_id1, style_id2, alpha) -> str:
    return "%s_%0.2f__%s" % (str(style_id1), alpha, str(style_id2))


class BrushLibrary:
    """``library.py:49-110``: ``from_arg`` accepts a file, ``rand<N>``, a seed count or a comma separated seed list."""

    @staticmethod
    def from_arg(arg_val: str, z_dim: int = 64) -> "BrushLibrary":
        if os.path.isfile(arg_val):
            return BrushLibrary.from_file(arg_val, z_dim=z_dim)
        m = re.match(r"^rand(\d+)$", arg_val)
        if m is not None:
            return RandomBrushLibrary(int(m.group(1)), zdim=z_dim)
        values = [int(x) for x in arg_val.split(",")]
        if len(values) == 1:
            seeds = list(range(0, max(10000, values[0])))
            random.shuffle(seeds)
            return SeedBrushLibrary(seeds[:values[0]], z_dim)
        return SeedBrushLibrary(values, z_dim)

    @staticmethod
    def from_file(fname: str, z_dim: int = 64) -> "BrushLibrary":
        try:
            return WBrushLibrary.from_file(fname, strict=True)
        except Exception:
            return SeedBrushLibrary.from_file(fname, z_dim=z_dim)

    def get_style_ids(self) -> List[str]:
        raise NotImplementedError

    def set_style(self, style_id, brush_options) -> None:
        raise NotImplementedError


class SeedBrushLibrary(BrushLibrary):
    @staticmethod
    def from_file(fname, z_dim=None):
        zs, zdim = read_zs(fname)
        return SeedBrushLibrary(zs, z_dim if z_dim is not None else zdim)

    def __init__(self, seeds_list, zdim):
        self.zs, self.zdim = list(seeds_list), zdim

    def get_style_ids(self):
        return sorted(str(x) for x in self.zs)

    def set_style(self, style_id, brush_options):
        z = torch.from_numpy(np.random.RandomState(seed=int(style_id)).randn(1, self.zdim))
        brush_options.set_style(z, style_id=style_id)

    def set_interpolated_style(self, style_id1, style_id2, alpha, brush_options):
        z1 = np.random.RandomState(seed=int(style_id1)).randn(1, self.zdim)
        z2 = np.random.RandomState(seed=int(style_id2)).randn(1, self.zdim)
        brush_options.set_style(torch.from_numpy(z1 * alpha + z2 * (1 - alpha)),
                                style_id=_interp_style_id(style_id1, style_id2, alpha))


class RandomBrushLibrary(BrushLibrary):
    """``rand<N>`` (library.py:237-251): ids ``rand0..rand<N-1>``; every ``set_style`` draws a fresh UNIFORM [0,1) latent from
    a torch generator seeded with 1 (``forger/metrics/util.py:77-89``, ``RandomState(0)``), whatever the id."""

    def __init__(self, num, zdim, seed=0):
        self.num, self.zdim = num, zdim
        self.tgenerator = torch.Generator()
        self.tgenerator.manual_seed(seed + 1)

    def get_style_ids(self):
        return ["rand" + str(x) for x in range(self.num)]

    def set_style(self, style_id, brush_options):
        brush_options.set_style(torch.rand((1, self.zdim), dtype=torch.float32, generator=self.tgenerator))

    def set_interpolated_style(self, style_id1, style_id2, alpha, brush_options):
        self.set_style(style_id1, brush_options)


class WBrushLibrary(BrushLibrary):
    """Projected brushes: pickle of {style_id: ws tensor | {'w': ws, 'noise': {name: tensor}} | {'w': ws, name: tensor…}}."""

    @staticmethod
    def from_file(fname, strict=False):
        styles = {}
        if os.path.isfile(fname):
            with open(fname, "rb") as f:
                styles = pickle.load(f)
            if not isinstance(styles, dict):
                raise ValueError("not a W library")
        elif strict:
            raise FileNotFoundError(fname)
        return WBrushLibrary(styles)

    def __init__(self, styles_dict):
        self.styles = styles_dict

    def get_style_ids(self):
        return sorted(self.styles.keys())

    @staticmethod
    def _split(style_info):
        if isinstance(style_info, dict):
            w = style_info["w"]
            noise = style_info["noise"] if "noise" in style_info else {k: v for k, v in style_info.items() if k != "w"}
            noise = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))) for k, v in noise.items()} or None
            return w, noise
        return style_info, None

    def set_style(self, style_id, brush_options):
        w, noise = self._split(self.styles[style_id])
        brush_options.set_style_w(w, style_id=style_id, custom_args={"noise_buffers": noise})

    def set_interpolated_style(self, style_id1, style_id2, alpha, brush_options):
        (w1, n1), (w2, n2) = self._split(self.styles[style_id1]), self._split(self.styles[style_id2])
        custom = None
        if n1 is not None and n2 is not None:
            custom = {"noise_buffers": {k: v * alpha + n2[k] * (1 - alpha) for k, v in n1.items()}}
        brush_options.set_style_w(w1 * alpha + w2 * (1 - alpha), style_id=_interp_style_id(style_id1, style_id2, alpha),
                                  custom_args=custom)


# ------------------------------------------------------------------------------------------------
# engine snapshot container (.npz)
# ------------------------------------------------------------------------------------------------
def save_engine_snapshot(path: str, cfg, generator_sd: Dict[str, np.ndarray], encoder_sd: Dict[str, np.ndarray],
                         preproc_type: Optional[str] = None, extra: Optional[dict] = None) -> None:
    """Generator tensors under their reference names, encoder tensors under ``encoder/<name>``, config + metadata JSON."""
    import dataclasses
    import json
    arrays = {k: np.asarray(v) for k, v in generator_sd.items()}
    arrays.update({"encoder/" + k: np.asarray(v) for k, v in encoder_sd.items()})
    meta = {"config": dataclasses.asdict(cfg), "preproc_type": preproc_type, "extra": extra or {}}
    arrays["__engine__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez(path, **arrays)


def load_engine_snapshot(path: str):
    """-> (GeneratorConfig, generator state dict, encoder state dict, preproc_type, extra)."""
    import json
    from .config import GeneratorConfig
    with np.load(path) as z:
        meta = json.loads(bytes(z["__engine__"]).decode())
        gen = {k: z[k] for k in z.files if not k.startswith("encoder/") and not k.startswith("__")}
        enc = {k[len("encoder/"):]: z[k] for k in z.files if k.startswith("encoder/")}
    c = meta["config"]
    for key in ("geom_feature_channels", "geom_feature_resolutions", "resample_filter"):
        if key in c and c[key] is not None:
            c[key] = tuple(c[key])
    return GeneratorConfig(**c), gen, enc, meta["preproc_type"], meta["extra"]
