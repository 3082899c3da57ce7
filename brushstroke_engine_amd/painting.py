"""Painting-engine driver on the MI355X generator (SURVEY 8 rows e / f2; BASELINE configs 1 and 3).

Counterpart of the reference's ``PaintingHelper.render_stroke`` + ``FeatureCanvas`` (``forger/ui/brush.py:33-92,
95-398``), ``TriadGanPaintEngine._render_stroke_torch`` (``brush.py:731-805``), ``generate_stitching_crops``
(``forger/viz/style_transfer.py:15-48``) and the tile loop of ``forger/viz/paint_image_main.py:30-63, 145-192``.

The reference paints a canvas tile by tile: with feature blending every tile reads what earlier tiles left on the
``FeatureCanvas``.  That dependency is pointwise at the blending resolution, so this build runs ALL tiles through a
three-phase schedule with identical results (verified against the reference-generated canvases):

  phase 1  blocks b4..b(R/2) of every tile, batched           (``Generator`` split entry ``_stop_after``)
  phase 2  one launch replays the canvas blend tile after tile (``nb_canvas_replay_f32``)
  phase 3  last block + triad ToRGB + compositing, batched     (``_resume``), then one paste launch

Multi-GPU: the tile list is cut into contiguous per-rank ranges (``sharding.shard_bounds``); all three phases run on
the rank's own tiles only.  The blend of a tile reads what EARLIER tiles left under it, so a rank needs from the ranks
before it just the strips of their tiles that overlap its own (``sharding.halo_plan``: 20 px wide at R/2 for crop
margin 10, ~0.65 MB per neighbour pair at R=256) -- ONE ``all_to_all_single`` of those strips over RCCL/xGMI, started
as soon as the rank's boundary tiles are through phase 1 (they are scheduled first) and overlapped with the rest of
phase 1.  Each rank then replays its own tiles plus the received strips (``nb_canvas_replay_pieces_f32``) on the cells
its tiles cover, and the RGBA tiles are gathered on rank 0 for the paste.

Device work goes through ``TileOps`` (HIP kernels + the PyTorch-ROCm encoder).  There is no CPU fallback here: the
tests inject an oracle-backed ``TileOps`` stand-in to check the host logic and the sharded schedule on CPU.
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .sharding import halo_plan, shard_bounds

CELL_H, CELL_W = 4, 64          # NB_CELL_H / NB_CELL_W of include/neube_hip.h
PAINT_SLOT0 = 8                 # first generator workspace slot of the tiled schedule (TileOps._GRAPH_SLOT = 16 onwards: graphs)


# ------------------------------------------------------------------------------------------------
# host-side geometry / tiling helpers (numpy; no device work)
# ------------------------------------------------------------------------------------------------
def threshold_otsu(image: np.ndarray) -> float:
    """Otsu threshold of an integer image over its occupied value range -- the algorithm of
    ``skimage.filters.threshold_otsu`` (scikit-image 0.19, one bin per integer value), which
    ``forger/util/img_proc.py:66-71`` calls.  scikit-image is not installed in the build image, so this helper is
    NOT pinned against it (DESIGN.md says so); it only prepares input geometry, before the measured path."""
    img = np.asarray(image)
    if img.size == 0:
        raise ValueError("empty image")
    lo, hi = int(img.min()), int(img.max())
    if lo == hi:
        return float(lo)
    counts = np.bincount(img.ravel().astype(np.int64) - lo, minlength=hi - lo + 1).astype(np.float64)
    centers = np.arange(lo, hi + 1, dtype=np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / np.maximum(w1, 1e-300)
    m2 = (np.cumsum((counts * centers)[::-1]) / np.maximum(w2[::-1], 1e-300))[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return float(centers[int(np.argmax(var12))])


def prepare_geometry_image(img: np.ndarray) -> np.ndarray:
    """``_read_any_geo`` (paint_image_main.py:30-55) from a decoded image array: gray / RGB / RGBA ->
    [H,W,1] uint8 with 255 = background, 0 = stroke (Otsu-thresholded)."""
    a = np.asarray(img).astype(np.float32)
    if a.ndim == 2:
        a = a[..., None]
    if a.shape[2] == 3:
        a = a[..., :3].mean(axis=2, dtype=np.float32)[..., None]
    elif a.shape[2] == 4:
        mean = a[..., :3].mean(axis=2, dtype=np.float32)
        alpha = a[..., 3] / np.float32(255)
        a = (mean * alpha + np.float32(255) * (1 - alpha))[..., None]
    mn = a.min()
    if mn > 0:
        a = a - mn
    mx = a.max()
    if 0 < mx < 255:
        a = a * np.float32(255.0 / float(mx))
    a8 = a.astype(np.uint8)
    return ((a8 > threshold_otsu(a8)).astype(np.float32) * 255).astype(np.uint8)


def read_geometry_image(fname: str) -> np.ndarray:
    from PIL import Image
    return prepare_geometry_image(np.array(Image.open(fname)))


def pad_geo(geo: np.ndarray, crop_margin: int) -> np.ndarray:
    """paint_image_main.py:58-61."""
    out = np.full((geo.shape[0] + crop_margin, geo.shape[1] + crop_margin, geo.shape[2]), 255, np.uint8)
    out[crop_margin:, crop_margin:, :] = geo
    return out


def generate_stitching_crops(stroke_image: np.ndarray, patch_width: int, mode: str = "all", overlap_margin: int = 15):
    """style_transfer.py:15-48: (y, x, P, P) crops at stride P - 2*overlap over the image padded with 255."""
    rwidth = patch_width - overlap_margin * 2
    assert rwidth > 0, "overlap margin too large for the patch width"
    h, w, ch = stroke_image.shape
    assert ch in (1, 2, 3, 4), f"Wrong shape {stroke_image.shape}"
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


def dirty_area_alpha(width: int, margin: int, crop_margin: int = 0) -> np.ndarray:
    """``PaintingHelper.generate_dirty_area_alpha`` (brush.py:159-187) for a dirty area spanning the whole tile:
    1 inside the rectangle inset by margin + crop_margin, linear fall-off of width ``margin`` outside (distance to the
    nearest edge; to the nearest corner in the corner regions), fp32 like the reference."""
    f32 = np.float32
    r0 = margin + crop_margin
    r1 = r0 + width - 2 * margin - 2 * crop_margin
    assert 0 <= r0 < r1 <= width, "blend margin + crop margin leave no interior"
    x = np.arange(width, dtype=f32)
    gy, gx = np.meshgrid(x, x, indexing="ij")
    dx = np.minimum((gx - f32(r0)) ** 2, (gx - f32(r1) + f32(1)) ** 2).astype(f32)
    dy = np.minimum((gy - f32(r0)) ** 2, (gy - f32(r1) + f32(1)) ** 2).astype(f32)
    d = (dx + dy).astype(f32)
    d[0:r0, r0:r1] = dy[0:r0, r0:r1]
    d[r1:, r0:r1] = dy[r1:, r0:r1]
    d[r0:r1, 0:r0] = dx[r0:r1, 0:r0]
    d[r0:r1, r1:] = dx[r0:r1, r1:]
    res = (f32(1) - np.sqrt(d).astype(f32) / f32(margin)).astype(f32)
    res[res < 0] = 0
    res[r0:r1, r0:r1] = 1
    return res


def build_cells(rects: np.ndarray, h: int, w: int) -> Tuple[np.ndarray, np.ndarray]:
    """CSR list of the rectangles (y0, x0, y1, x1; end-exclusive) touching each CELL_H x CELL_W cell of an h x w
    grid, in ascending rectangle order (include/neube_hip.h, "Cells").  Vectorised: canvases have thousands of tiles."""
    ncx, ncy = -(-w // CELL_W), -(-h // CELL_H)
    r = np.asarray(rects, np.int64).reshape(-1, 4)
    y0, x0 = np.maximum(r[:, 0], 0), np.maximum(r[:, 1], 0)
    y1, x1 = np.minimum(r[:, 2], h), np.minimum(r[:, 3], w)
    ok = (y1 > y0) & (x1 > x0)
    off = np.zeros(ncx * ncy + 1, np.int32)
    if not ok.any():
        return off, np.zeros(1, np.int32)
    t_idx = np.nonzero(ok)[0]
    cy0, cx0 = y0[ok] // CELL_H, x0[ok] // CELL_W
    ny, nx = (y1[ok] - 1) // CELL_H - cy0 + 1, (x1[ok] - 1) // CELL_W - cx0 + 1
    cnt = ny * nx
    rep = np.repeat(np.arange(len(t_idx)), cnt)                       # which rectangle each (cell, tile) pair belongs to
    local = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    cells = (cy0[rep] + local // nx[rep]) * ncx + cx0[rep] + local % nx[rep]
    tiles = t_idx[rep]
    order = np.lexsort((tiles, cells))                                # by cell, then ascending tile index
    np.cumsum(np.bincount(cells, minlength=ncx * ncy), out=off[1:])
    return off, tiles[order].astype(np.int32)


# ------------------------------------------------------------------------------------------------
# brush options (the subset of GanBrushOptions, brush.py:410-527, that the tiled path reads)
# ------------------------------------------------------------------------------------------------
class GanBrushOptions:
    def __init__(self, primary_color=None, secondary_color=None):
        self.color0 = self.color1 = self.canvas_color = None
        self.style_z = self.style_ws = self.style_id = None
        self.position = None
        self.custom_args: Dict = {}
        self.enable_uvs_mapping = False
        if primary_color is not None:
            self.set_color(0, primary_color)
        if secondary_color is not None:
            self.set_color(1, secondary_color)

    def set_position(self, x, y):
        self.position = torch.tensor([[int(y), int(x)]], dtype=torch.int64)

    def set_color(self, color_idx: int, in_color):
        c = None
        if in_color is not None:
            c = torch.as_tensor(np.asarray(in_color))
            c = c.to(torch.float32) / 255 if c.dtype == torch.uint8 else c.to(torch.float32)
            c = c.reshape(-1, 3)
        if color_idx == 0:
            self.color0 = c
        elif color_idx == 1:
            self.color1 = c
        elif color_idx == 2:
            self.canvas_color = c
        else:
            raise RuntimeError(f"Wrong color idx {color_idx}")

    def set_style(self, style_z, style_id=None):
        self.style_z, self.style_id, self.style_ws = style_z, style_id, None

    def set_style_w(self, style_w, style_id=None, custom_args=None):
        self.style_ws, self.style_id, self.style_z = style_w, style_id, None
        self.custom_args = {} if custom_args is None else custom_args

    def user_colors(self) -> Optional[torch.Tensor]:
        """[1,3(rgb),3(k)] with NaN = keep the generator's color (``prepare_colors``, brush.py:514-527)."""
        if self.color0 is None and self.color1 is None and self.canvas_color is None:
            return None
        uc = torch.full([1, 3, 3], float("nan"), dtype=torch.float32)
        for k, c in enumerate((self.color0, self.color1, self.canvas_color)):
            if c is not None:
                uc[0, :, k] = c[0]
        return uc


# ------------------------------------------------------------------------------------------------
# device operations (HIP)
# ------------------------------------------------------------------------------------------------
def _p(t):
    return None if t is None else t.data_ptr()


class TileOps:
    """What the tiled schedule needs from the device: the HIP generator split in two, the geometry encoder, and the
    canvas kernels of ``csrc/nb_canvas.hip``."""

    def __init__(self, G, encoder, device=None):
        self.G, self.encoder = G, encoder
        self.device = torch.device(device) if device is not None else G.synthesis.get_last_block().conv1.weight.device
        if self.device.type != "cuda":
            raise _lib.NeubeHipError("the painting engine needs the generator on a GPU (no CPU path in this build)")
        self.patch_width = G.img_resolution
        self.cfg = G.cfg
        # the encoder computes in the generator's arithmetic: fp8 correction operands between its layers only when the generator
        # itself runs "f8" (its features then carry ~3e-4 instead of ~2e-5 absolute error on O(5) values)
        if hasattr(encoder, "arith"):
            encoder.arith = "f8" if getattr(G.synthesis, "conv_mode", None) == "f8" else "h3"
        self._rr = {}                # stream count -> pipeline.RoundRobinStreams (created at first use, kept across probes)
        self._graphs, self._graph_epoch, self._capturing = {}, None, False

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- inputs --
    def to_device(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True)

    def to_host(self, t: torch.Tensor) -> np.ndarray:
        """Device tensor -> numpy through a pinned buffer (the caching host allocator recycles it once the array dies)."""
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return host.numpy()

    def geom_tiles(self, geom_dev: torch.Tensor, tile_yx: torch.Tensor) -> torch.Tensor:
        gh, gw = geom_dev.shape
        t, r = tile_yx.shape[0], self.patch_width
        out = torch.empty([t, 1, r, r], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().nb_geom_tiles_f32(_p(geom_dev), gh, gw, _p(tile_yx), t, r, _p(out), self._stream()),
                       "geom_tiles")
        return out

    lazy_geometry = True       # hand the generator a geometry provider (encoder output straight into the layer operands)

    def encode(self, geom: torch.Tensor):
        if self.lazy_geometry and hasattr(self.encoder, "lazy"):
            return self.encoder.lazy(geom)
        return self.encoder.encode(geom)

    def prepare(self, n: int, slots) -> None:
        """Create the generator workspaces of ``slots`` for batches of ``n`` on the CALLER's stream, before the batch streams fork
        (a workspace is otherwise created by the first batch that needs it, on that batch's stream, while the other stream is
        already running)."""
        syn = self.G.synthesis
        syn._n, syn._h3_batch_ok = n, n >= syn.h3_min_batch
        for sl in slots:
            syn._get_plan(n, self.device, sl)

    def map_style(self, z=None, ws=None) -> torch.Tensor:
        if ws is not None:
            return ws.to(self.device, torch.float32)
        return self.G.mapping(z.to(self.device), None)

    # -- generator halves --
    # ``slot``: which of the generator's per-batch workspaces to use -- consecutive batches alternate between two HIP
    # streams (``stream``/``join_streams``), each with its own workspace, so that one batch's kernel tails and small
    # launches are covered by the other batch's work
    def head(self, ws, geom_feats, positions, stop_res: int, slot: int = 0) -> torch.Tensor:
        if self._use_graph(ws):
            return self._graphed(("head", stop_res), [ws, geom_feats[0], geom_feats[1], positions],
                                 lambda w, g0, g1, ps: self.head(w, [g0, g1], ps, stop_res, slot=self._GRAPH_SLOT))
        return self.G.forward_pre_mapped(ws, geom_feats, positions=positions, noise_mode="const", _stop_after=stop_res,
                                         _plan_slot=slot)

    def tail(self, ws, feats, geom_feats, positions, resume_res: int, render_mode, user_colors, sfactor=None,
             slot: int = 0) -> torch.Tensor:
        if self._use_graph(ws) and (sfactor is None or torch.is_tensor(sfactor)):
            return self._graphed(("tail", resume_res, render_mode), [ws, feats, geom_feats[0], geom_feats[1], positions, user_colors, sfactor],
                                 lambda w, f, g0, g1, ps, uc, sf: self.tail(w, f, [g0, g1], ps, resume_res, render_mode, uc, sf,
                                                                           slot=self._GRAPH_SLOT + 1))
        u8, _, _ = self.G.render_triad(ws=ws, geom_feature=geom_feats, positions=positions, render_mode=render_mode,
                                       user_colors=user_colors, sfactor=sfactor, _resume=(resume_res, feats), _plan_slot=slot)
        return u8

    def full(self, ws, geom_feats, positions, render_mode, user_colors, sfactor=None, slot: int = 0) -> torch.Tensor:
        if self._use_graph(ws) and (sfactor is None or torch.is_tensor(sfactor)):
            return self._graphed(("full", render_mode), [ws, geom_feats[0], geom_feats[1], positions, user_colors, sfactor],
                                 lambda w, g0, g1, ps, uc, sf: self.full(w, [g0, g1], ps, render_mode, uc, sf,
                                                                         slot=self._GRAPH_SLOT + 2))
        u8, _, _ = self.G.render_triad(ws=ws, geom_feature=geom_feats, positions=positions, render_mode=render_mode,
                                       user_colors=user_colors, sfactor=sfactor, _plan_slot=slot)
        return u8

    # -- hipGraph replay of the generator passes of ONE tile (interactive strokes) --
    # An interactive stroke is bound by the ~0.2 ms of Python each eager generator pass costs, not by the device.  With
    # ``graph_single`` set (PaintingHelper.render_stroke does), the passes of a single tile are captured once per call
    # signature into hipGraphs that own a workspace slot; a call then copies its inputs into the graph's static
    # tensors and replays.  The returned tensor is the graph's output buffer: valid until the next call of that pass.
    graph_single = False
    _GRAPH_SLOT = 16

    def _use_graph(self, ws) -> bool:
        return (self.graph_single and not self._capturing and ws.shape[0] == 1 and len(self.cfg.geom_feature_channels) == 2
                and not torch.cuda.is_current_stream_capturing())

    def _graphed(self, key, tensors, fn):
        packed = self.G.synthesis.packed
        if self._graph_epoch is not packed:                   # weights were (re)loaded / moved: captured pointers are stale
            self._graphs, self._graph_epoch = {}, packed
        key = key + tuple(None if t is None else (tuple(t.shape), t.dtype) for t in tensors)
        ent = self._graphs.get(key)
        if ent is None:
            statics = [None if t is None else t.detach().clone() for t in tensors]
            self._capturing = True
            try:
                cur = torch.cuda.current_stream(self.device)
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):                # creates the slot's workspace, packs weights, warms the kernels
                    for _ in range(2):
                        fn(*statics)
                cur.wait_stream(side)
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = fn(*statics)
            finally:
                self._capturing = False
            ent = (graph, statics, out)
            self._graphs[key] = ent
        graph, statics, out = ent
        for s_, t in zip(statics, tensors):
            if s_ is not None:
                s_.copy_(t, non_blocking=True)
        graph.replay()
        return out

    n_streams = 2
    # 0 = choose n_streams by a probe the first time a multi-batch canvas is painted (choose_streams); 1 / 2 / 3 = fixed.
    # Alternating batches over HIP streams lets one batch's kernel tails and small launches run under the other's big ones, which
    # is worth +5..10 % on most boxes -- and cost 14 % on the box of the round-3 driver run (three chains of 156 KB-LDS workgroups
    # evicting each other at kernel boundaries), so the schedule measures instead of assuming.
    stream_policy = int(os.environ.get("NB_CANVAS_STREAMS", "0"))
    stream_probe = None          # what choose_streams measured: {"ms_per_batch": {1: .., 2: ..}, "chosen": n, "batch": n}

    def _set_n_streams(self, k: int) -> None:
        """Change the number of batch streams (every count keeps its own ``RoundRobinStreams``; ``stream(k)`` indexes the current one)."""
        self.n_streams = k

    def _round_robin(self):
        """The scheduler of the generator's throughput mode (``pipeline.RoundRobinStreams``, also behind ``ConcurrentTriadSteps`` /
        ``bench.py``): batch b runs on stream b % n_streams with workspace slot PAINT_SLOT0 + b % n_streams."""
        from .pipeline import RoundRobinStreams
        rr = self._rr.get(self.n_streams)
        if rr is None:
            rr = self._rr[self.n_streams] = RoundRobinStreams(self.device, self.n_streams, PAINT_SLOT0)
        return rr

    def choose_streams(self, n: int, render_mode: str = "clear") -> int:
        """Pick the number of batch streams for batches of ``n`` tiles: the fixed policy, or -- once per TileOps and batch size --
        two streams unless a probe -- three interleaved rounds of six synthetic batches on 1 and on 2 streams, best of three each, full
        generator passes with random styles and geometry features, ~80 ms -- shows them more than 3 % slower.  Sets ``n_streams``; the probe's figures stay in ``stream_probe`` (tools/bench_canvas.py reports them)."""
        if self.stream_policy > 0:
            self._set_n_streams(self.stream_policy)
            return self.n_streams
        if self.stream_probe is not None and self.stream_probe.get("batch") == n:
            return self.n_streams
        cfg, dev = self.cfg, self.device
        gen = torch.Generator(device="cpu").manual_seed(1234)
        ws = torch.randn([n, cfg.num_ws, cfg.w_dim], generator=gen).to(dev)
        geom = [torch.randn([n, c, r, r], generator=gen).to(dev) for c, r in zip(cfg.geom_feature_channels, cfg.geom_feature_resolutions)]
        pos = torch.zeros([n, 2], dtype=torch.int64, device=dev)
        times = {1: [], 2: []}

        def run(k, batches):
            # (each stream count keeps its own side streams across the interleaved rounds)
            self.n_streams = k
            self.prepare(n, sorted({PAINT_SLOT0 + i % k for i in range(k)}))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            outs = []
            for b in range(batches):
                with self.stream(b):
                    outs.append(self.full(ws, geom, pos, render_mode, None, None, slot=PAINT_SLOT0 + b % k))
            self.join_streams(outs)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / batches * 1e3
        run(1, 2); run(2, 4)                             # workspaces, code objects, clocks
        for _ in range(3):                               # interleaved, best of three each: a single pair of short runs was off by
            times[1].append(run(1, 6))                   # +-7 % between processes on one box (r04 collection: 1.89 vs 2.02 ms, then
            times[2].append(run(2, 6))                   # 1.87 vs 1.77 ms a minute later), more than the effect it is meant to see
        times = {k: min(v) for k, v in times.items()}
        # two streams unless they are clearly slower HERE.  The probe renders generator passes only; in the canvas job, where encoder,
        # canvas kernels and copies sit between them, the second stream is worth 5-7 % on every box measured (same-box A/Bs of the
        # 4096^2 job in round 4: 44.4 -> 42.2, 44.1 -> 41.0, 43.8 -> 40.8 ms) while their probes said 1.81 -> 1.79, 2.08 -> 1.81 and
        # 1.85 -> 1.87 ms per batch: a probe within noise must not cost the job its overlap; a box that loses with concurrent chains
        # (the round-3 driver run saw three streams 14 % below one) shows it as > 3 % here
        best = 1 if times[2] > 1.03 * times[1] else 2
        self.n_streams = best
        self.stream_probe = {"ms_per_batch": {str(k): round(v, 4) for k, v in times.items()}, "chosen": best, "batch": n}
        return best

    def stream(self, k: int):
        """Context manager: work of batch k goes to side stream k % n_streams (which first waits for the caller's)."""
        return self._round_robin().stream(k)

    def join_streams(self, tensors=()):
        """The caller's stream waits for the side streams; ``tensors`` produced there are about to be read here."""
        rr = self._rr.get(self.n_streams)
        if rr is not None:
            rr.join(tensors)
        else:
            main = torch.cuda.current_stream(self.device)
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)

    def comm_stream(self):
        """Context manager for the halo exchange: a side stream that first waits for everything enqueued so far on the
        batch streams (the boundary tiles' phase 1), so that packing + the collective overlap the remaining batches."""
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream(device=self.device)
        self._comm.wait_stream(torch.cuda.current_stream(self.device))
        rr = self._rr.get(self.n_streams)
        for st in (rr.streams if rr is not None else []):
            self._comm.wait_stream(st)
        return torch.cuda.stream(self._comm)

    def join_comm(self, tensors=()):
        if getattr(self, "_comm", None) is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self._comm)
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)

    def background_weight(self, ws, geom_feats) -> torch.Tensor:
        """S = uvs[:, 2:3] of an un-positioned render (what StyleUVSMapper calibrates on, mapper.py:74-87)."""
        _, dbg = self.G.forward_pre_mapped(ws, geom_feats, return_debug_data=True, noise_mode="const")
        return dbg["uvs"][:, 2:3]

    # -- canvas kernels --
    def new_feature_canvas(self, c, hc, wc):
        return (torch.zeros([1, c, hc, wc], dtype=torch.float32, device=self.device),
                torch.zeros([hc, wc], dtype=torch.uint8, device=self.device))

    def replay(self, tiles, tile_yx, alpha0, crop, canvas, mask, cell_off, cell_tiles, box=None) -> torch.Tensor:
        """In place on ``tiles`` and ``canvas``; returns the new mask buffer.  ``box`` = (y0, x0, y1, x1) canvas pixels
        that contain every tile (an interactive stroke replays one tile on a large canvas): only those cells run."""
        t, c, hw, _ = tiles.shape
        hc, wc = mask.shape
        with torch.cuda.device(self.device):
            if box is not None:
                y0, x0 = max(0, box[0]) // CELL_H, max(0, box[1]) // CELL_W
                y1, x1 = -(-min(hc, box[2]) // CELL_H), -(-min(wc, box[3]) // CELL_W)
                if y1 <= y0 or x1 <= x0:
                    return mask
                mask_out = mask.clone()
                _lib.check(_lib.lib().nb_canvas_replay_box_f32(_p(tiles), t, c, hw, _p(tile_yx), _p(alpha0), crop, _p(canvas),
                                                               _p(mask), _p(mask_out), hc, wc, _p(cell_off), _p(cell_tiles),
                                                               x0, y0, x1 - x0, y1 - y0, self._stream()), "canvas_replay")
                return mask_out
            mask_out = torch.empty_like(mask)
            _lib.check(_lib.lib().nb_canvas_replay_f32(_p(tiles), t, c, hw, _p(tile_yx), _p(alpha0), crop, _p(canvas),
                                                       _p(mask), _p(mask_out), hc, wc, _p(cell_off), _p(cell_tiles),
                                                       self._stream()), "canvas_replay")
        return mask_out

    def replay_pieces(self, pieces, hw, alpha0, crop, canvas, mask, cell_off, cell_pieces, box) -> torch.Tensor:
        """The replay over tile pieces (multi-GPU halo exchange).  ``pieces`` = [(view [C,h,w] with unit column stride,
        cy, cx, ly0, lx0), ...] in paint order; in place on the views and ``canvas``; returns the new mask buffer.
        ``box`` = (y0, x0, y1, x1) canvas pixels containing every piece."""
        import ctypes as C
        hc, wc = mask.shape
        c = pieces[0][0].shape[0]
        arr = (_lib.NbTilePiece * len(pieces))()
        for i, (v, cy, cx, ly0, lx0) in enumerate(pieces):
            assert v.dtype == torch.float32 and v.stride(2) == 1 and v.shape[0] == c
            arr[i] = _lib.NbTilePiece(v.data_ptr(), v.stride(0), v.stride(1), cy, cx, v.shape[1], v.shape[2], ly0, lx0)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device, non_blocking=True)
        y0, x0 = max(0, box[0]) // CELL_H, max(0, box[1]) // CELL_W
        y1, x1 = -(-min(hc, box[2]) // CELL_H), -(-min(wc, box[3]) // CELL_W)
        if y1 <= y0 or x1 <= x0:
            return mask
        mask_out = mask.clone()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().nb_canvas_replay_pieces_f32(_p(table), len(pieces), c, hw, _p(alpha0), crop, _p(canvas),
                                                              _p(mask), _p(mask_out), hc, wc, _p(cell_off), _p(cell_pieces),
                                                              x0, y0, x1 - x0, y1 - y0, self._stream()), "canvas_replay_pieces")
        return mask_out

    def paste(self, canvas_u8, tiles_u8, dst_yx, crop, cell_off, cell_tiles) -> None:
        t, r = tiles_u8.shape[0], tiles_u8.shape[1]
        h, w = canvas_u8.shape[:2]
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().nb_paste_tiles_u8(_p(tiles_u8), t, r, _p(dst_yx), crop, _p(canvas_u8), h, w,
                                                    _p(cell_off), _p(cell_tiles), self._stream()), "paste_tiles")


# ------------------------------------------------------------------------------------------------
# clear-background mapping (reference: StyleUVSMapper, forger/ui/mapper.py:16-72, 117-135)
# ------------------------------------------------------------------------------------------------
class StyleUVSMapper:
    """Per-style scale ``sfactor`` that stretches the background weight S of the triad so that clear background
    renders fully transparent; the remap itself (``_map_style_s``) is fused into the ToRGB launch
    (``nb_torgb_triad_f32(sfactor=...)``).  The reference calibrates on five bundled drawings at two stroke widths
    (``*_rad016.png`` / ``*_rad025.png``); they are data files of the reference, so the caller supplies them."""

    def __init__(self, ops, cal_medium: Optional[np.ndarray] = None, cal_thick: Optional[np.ndarray] = None):
        self.ops = ops
        self.sfactors: Dict = {}
        self.geom_feature = self.bmask = None
        if cal_medium is not None:
            self.set_calibration(cal_medium, cal_thick)

    def set_calibration(self, cal_medium: np.ndarray, cal_thick: np.ndarray):
        """[k,R,R] uint8 drawings, 0 = stroke: medium strokes are rendered, thick ones define "surely background"."""
        geo = (self.ops.to_device(np.asarray(cal_medium, np.uint8)).to(torch.float32) / 255).unsqueeze(1)
        self.geom_feature = self.ops.encode(geo)
        self.bmask = (self.ops.to_device(np.asarray(cal_thick, np.uint8)).to(torch.float32) / 255).unsqueeze(1) > 0.99
        self.sfactors = {}

    def get_sfactor(self, opts) -> torch.Tensor:
        if opts.style_id is not None and opts.style_id in self.sfactors:
            return self.sfactors[opts.style_id]
        if self.geom_feature is None:
            raise RuntimeError("StyleUVSMapper: no calibration drawings set (set_calibration)")
        n = self.geom_feature[0].shape[0]
        ws = self.ops.map_style(z=opts.style_z, ws=opts.style_ws).expand(n, -1, -1).contiguous()
        S = self.ops.background_weight(ws, self.geom_feature)
        val = torch.stack([torch.topk(S[i][self.bmask[i]], k=15)[0].min() for i in range(n)]).min()
        sfactor = 1 / val
        if opts.style_id is not None:
            self.sfactors[opts.style_id] = sfactor
        return sfactor


# ------------------------------------------------------------------------------------------------
# the painting helper
# ------------------------------------------------------------------------------------------------
class PaintingHelper:
    """Server-side canvas state + rendering (reference: ``PaintingHelper``, brush.py:95-398).

    ``render_stroke`` keeps the reference's one-tile-per-call contract (interactive strokes); ``render_tiles`` takes a
    whole tile list through the three-phase schedule and is what ``paint_image`` uses."""

    feature_blending_margin = 16

    def __init__(self, ops: TileOps, batch: int = 32, group=None, uvs_mapper: Optional[StyleUVSMapper] = None):
        self.ops = ops
        self.uvs_mapper = uvs_mapper
        self.batch = int(batch)
        self.group = group
        self.patch_width = ops.patch_width
        self.render_mode = "clear"
        self.feature_blending_level = 0
        self.rows = self.cols = None
        self.features = self.mask = None
        self._canvas_rank_local, self._sync_layout = False, None
        self.halo_bytes = None
        self.down_factor = None
        self._alpha_cache: Dict[Tuple[int, int, int], torch.Tensor] = {}

    # -- canvas state --
    def make_new_canvas(self, rows: int, cols: int, feature_blending: Optional[int] = None):
        self.rows, self.cols = int(rows), int(cols)
        self.set_feature_blending(self.feature_blending_level if feature_blending is None else feature_blending)

    def set_feature_blending(self, feature_blending_level: int = 0):
        self.feature_blending_level = int(feature_blending_level)
        self.features = self.mask = None
        self._canvas_rank_local, self._sync_layout = False, None
        self.down_factor = 2 ** (self.feature_blending_level - 1) if self.feature_blending_level > 0 else None

    def sync_canvas(self):
        """Make the feature canvas whole on every rank after a sharded ``render_tiles`` (collective: all ranks call it).

        A sharded call replays, on each rank, the paint sequence under that rank's own tiles only (plus the strips of
        earlier foreign tiles under them), so a rank's canvas is final exactly where the LAST tile covering a pixel is one
        of its own.  Every pixel under this call's tiles therefore has one owning rank; the owners' values are summed with
        zeros from everybody else (one all-reduce over the tiles' bounding box: exact, x + 0 + ... + 0) and every rank ends
        up with the canvas the reference's single persistent ``FeatureCanvas`` (brush.py:33-92) would hold.  Runs lazily --
        ``_schedule`` calls it when a second sharded call paints on the same canvas; a canvas that is painted once (the
        ``paint_image`` job) never pays for it."""
        if not self._canvas_rank_local:
            return
        rank, world = self._world()
        rects, bounds = self._sync_layout
        self._canvas_rank_local, self._sync_layout = False, None
        if not self._sharded(world) or self.features is None:
            return
        hc, wc = self.mask.shape
        y0, x0 = max(0, int(rects[:, 0].min())), max(0, int(rects[:, 1].min()))
        y1, x1 = min(hc, int(rects[:, 2].max())), min(wc, int(rects[:, 3].max()))
        if y1 <= y0 or x1 <= x0:
            return
        last = np.full((hc, wc), -1, np.int64)                       # paint order: the last tile over a pixel wins
        owner_of = np.empty(rects.shape[0] + 1, np.int64)
        owner_of[-1] = -1
        for r_, (a, b) in enumerate(bounds):
            owner_of[a:b] = r_
        for t, (a0, b0, a1, b1) in enumerate(rects.tolist()):
            last[max(0, a0):a1, max(0, b0):b1] = t
        owner = owner_of[last[y0:y1, x0:x1]]
        ops = self.ops
        mine = ops.to_device((owner == rank))
        touched = ops.to_device((owner >= 0))
        f = self.features[0, :, y0:y1, x0:x1]
        m = self.mask[y0:y1, x0:x1]
        buf = torch.cat([(f * mine).reshape(-1), (m * mine).to(torch.float32).reshape(-1)])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        nf = f.numel()
        self.features[0, :, y0:y1, x0:x1] = torch.where(touched, buf[:nf].view(f.shape), f)
        self.mask[y0:y1, x0:x1] = torch.where(touched, buf[nf:].view(m.shape).to(torch.uint8), m)

    def set_render_mode(self, mode: str):
        if mode not in ("clear", "full"):
            raise RuntimeError("Unknown render mode for TriadGanPaintEngine: {}".format(mode))
        self.render_mode = mode

    def _alpha0(self, width, margin, crop) -> torch.Tensor:
        key = (width, margin, crop)
        if key not in self._alpha_cache:
            self._alpha_cache[key] = self.ops.to_device(dirty_area_alpha(width, margin, crop))
        return self._alpha_cache[key]

    # -- world --
    def _world(self) -> Tuple[int, int]:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _sharded(self, world: int) -> bool:
        """Does the call take the multi-rank schedule (halo exchange, pieces replay, tile gather)?  With more than one rank -- or at
        world size 1 under ``NB_FORCE_PG=1`` once the process group exists (launch.py: the very collectives of the N > 1 job, through
        RCCL, on a one-GPU box; same canvases as the single-process schedule)."""
        return world > 1 or (os.environ.get("NB_FORCE_PG") == "1" and dist.is_available() and dist.is_initialized())

    # -- the schedule --
    def _schedule(self, geom_img: np.ndarray, geom_yx: np.ndarray, areas_yx: np.ndarray, positions: Optional[np.ndarray],
                  opts: GanBrushOptions, crop_margin: int) -> Optional[torch.Tensor]:
        """Tiles i = 0..T-1 in paint order: geometry cut from ``geom_img`` [H,W] uint8 (255 = background) at
        ``geom_yx[i]``, canvas area at ``areas_yx[i]`` (already floored to the blending grid), noise position
        ``positions[i]`` (None: unshifted noise).  Returns the RGBA tiles [T,R,R,4] uint8 on rank 0 (None elsewhere)
        and leaves the feature canvas updated -- the result of the reference's tile-by-tile loop."""
        ops, R = self.ops, self.patch_width
        rank, world = self._world()
        sharded = self._sharded(world)
        T = areas_yx.shape[0]
        level, df = self.feature_blending_level, self.down_factor
        if sharded and level > 0 and self._canvas_rank_local:
            self.sync_canvas()                   # the previous sharded call left every rank with ITS tiles' paint sequence only
        t0, t1 = shard_bounds(T, rank, world)
        n_own = t1 - t0
        n_pad = -(-T // world)                                                 # equal per-rank count for collectives
        counts = [shard_bounds(T, r, world)[1] - shard_bounds(T, r, world)[0] for r in range(world)]

        geom_dev = ops.to_device(np.ascontiguousarray(geom_img))
        ws1 = ops.map_style(z=opts.style_z, ws=opts.style_ws)                  # one brush style for all tiles
        user = opts.user_colors()
        sfac = None
        if opts.enable_uvs_mapping:                                            # brush.py:773-774
            if self.uvs_mapper is None:
                raise RuntimeError("opts.enable_uvs_mapping needs a StyleUVSMapper (PaintingHelper(uvs_mapper=...))")
            sfac = self.uvs_mapper.get_sfactor(opts)
        own_yx = ops.to_device(geom_yx[t0:t1].astype(np.int32)) if n_own else None
        own_pos = ops.to_device(positions[t0:t1].astype(np.int64)) if (positions is not None and n_own) else None

        def batches():
            for b0 in range(0, n_own, self.batch):
                yield b0, min(b0 + self.batch, n_own)

        def style(n):
            return ws1.expand(n, -1, -1).contiguous()

        def pos(b0, b1):
            return None if own_pos is None else own_pos[b0:b1]

        on_stream = getattr(ops, "stream", lambda k: contextlib.nullcontext())      # (CPU stand-ins have no streams)
        join = getattr(ops, "join_streams", lambda tensors=(): None)
        # workspace slots of the painting schedule: disjoint from the generator's own sub-batch slots (1..sub_streams), whose
        # side-stream kernels of an un-joined throughput call may still be reading theirs
        # 1 or 2 batch streams: measured, once per TileOps -- by jobs long enough for the ~80 ms probe to be noise (smaller ones keep the
        # default of two: what they could gain or lose is a fraction of a millisecond)
        if hasattr(ops, "choose_streams") and (n_own >= 6 * self.batch or getattr(ops, "stream_policy", 0) > 0):
            ops.choose_streams(min(self.batch, n_own), self.render_mode)          # (a fixed policy applies to every canvas size)
        plan_slot = lambda k: PAINT_SLOT0 + k % getattr(ops, "n_streams", 1)
        if n_own <= self.batch:                      # a single batch (interactive strokes): nothing to overlap with
            on_stream, join, plan_slot = (lambda k: contextlib.nullcontext()), (lambda tensors=(): None), (lambda k: 0)
        elif hasattr(ops, "prepare"):
            ops.prepare(min(self.batch, n_own), sorted({plan_slot(k) for k in range(getattr(ops, "n_streams", 1))}))
        user_dev = None if user is None else user.to(ops.device)
        colors = lambda n_: None if user_dev is None else user_dev.expand(n_, -1, -1).contiguous()
        outs = []
        if level == 0:
            for k, (b0, b1) in enumerate(batches()):
                with on_stream(k):
                    g = ops.geom_tiles(geom_dev, own_yx[b0:b1])
                    outs.append(ops.full(style(b1 - b0), ops.encode(g), pos(b0, b1), self.render_mode, colors(b1 - b0), sfac,
                                         slot=plan_slot(k)))
            join(outs)
        else:
            bres = R // df
            C = ops.cfg.channels(bres)
            margin, crop_sc = self.feature_blending_margin // df, crop_margin // df
            rects = np.concatenate([areas_yx // df, areas_yx // df + bres], axis=1).astype(np.int64)
            if self.features is None:
                self.features, self.mask = ops.new_feature_canvas(C, -(-self.rows // df), -(-self.cols // df))
            hc, wc = self.mask.shape
            alpha0 = self._alpha0(bres, margin, crop_sc)
            # halo plan: the strips of earlier foreign tiles under my tiles (recv) and of my tiles under later ranks' (send)
            bounds = [shard_bounds(T, r, world) for r in range(world)]
            plan = halo_plan(rects, bounds) if sharded else {}
            send = {d: plan[(rank, d)] for d in range(world) if (rank, d) in plan}
            recv = {s_: plan[(s_, rank)] for s_ in range(world) if (s_, rank) in plan}
            # one rank under NB_FORCE_PG=1: nothing to exchange, so the all-to-all would carry no bytes -- let it carry ONE strip of
            # my first tile to myself (checked bit for bit after the exchange, not replayed): the packing, the split lists and the
            # collective itself then run as they do between ranks
            self_probe = None
            if sharded and world == 1 and n_own:
                r0 = rects[t0]
                self_probe = (t0, (int(r0[0]), int(r0[1]), int(r0[0]) + min(bres, 20), int(r0[3])))
                send = {0: [self_probe]}
                recv_probe = {0: [self_probe]}
            first_send = min((f for lst in send.values() for f, _ in lst), default=t1) - t0
            # phase 1: everything up to the blending resolution, own tiles.  The batches run last-to-first: the tiles the
            # later ranks need (the end of my range) finish first, and their strips travel under the rest of phase 1.
            mine = torch.empty([n_own, C, bres, bres], dtype=torch.float32, device=ops.device)
            geom_feats_own = {}
            work = recv_buf = send_buf = None
            nfl = lambda lst: sum(C * (q[2] - q[0]) * (q[3] - q[1]) for _, q in lst)
            in_split = [nfl(send.get(d, [])) for d in range(world)]
            out_split = [nfl((recv if self_probe is None else recv_probe).get(s_, [])) for s_ in range(world)]
            comm = getattr(ops, "comm_stream", lambda: contextlib.nullcontext())

            def exchange():
                nonlocal work, recv_buf, send_buf
                recv_buf = torch.empty([sum(out_split)], dtype=torch.float32, device=ops.device)
                with comm():
                    parts = [mine[f - t0, :, q[0] - rects[f, 0]:q[2] - rects[f, 0], q[1] - rects[f, 1]:q[3] - rects[f, 1]].reshape(-1)
                             for d in range(world) for f, q in send.get(d, [])]
                    send_buf = torch.cat(parts) if parts else torch.empty([0], dtype=torch.float32, device=ops.device)
                    work = dist.all_to_all_single(recv_buf, send_buf, out_split, in_split, group=self.group, async_op=True)
                self.halo_bytes = {"sent": 4 * sum(in_split), "received": 4 * sum(out_split)}

            blist = list(enumerate(batches()))
            for k, (b0, b1) in reversed(blist):
                with on_stream(k):
                    gf = ops.encode(ops.geom_tiles(geom_dev, own_yx[b0:b1]))
                    geom_feats_own[k] = gf
                    mine[b0:b1] = ops.head(style(b1 - b0), gf, pos(b0, b1), bres, slot=plan_slot(k))
                if sharded and work is None and b0 <= first_send:
                    exchange()
            if sharded and work is None:
                exchange()                                                        # (a rank without tiles still takes part)
            join()
            if work is not None:
                # (events on the caller's stream: the first fires when phase 1 is through, the second when the strips are there
                #  -- what of the exchange was NOT hidden under phase 1; comm_times())
                ev = self._comm_event_pair("halo_exchange_exposed_ms")
                work.wait()
                getattr(ops, "join_comm", lambda tensors=(): None)([recv_buf])
                if ev is not None:
                    ev[1].record()
                if self_probe is not None:
                    f, q = self_probe
                    src = mine[f - t0, :, q[0] - rects[f, 0]:q[2] - rects[f, 0], q[1] - rects[f, 1]:q[3] - rects[f, 1]].reshape(-1)
                    if recv_buf.numel() != src.numel() or not torch.equal(recv_buf, src):
                        raise RuntimeError("NB_FORCE_PG: the strip sent to myself through all_to_all_single came back changed")
            # phase 2: the sequential canvas blend of my tiles, replayed in one launch
            if not sharded:
                off, lst = build_cells(rects, hc, wc)
                # few tiles on a large canvas (interactive strokes): replay only the cells inside their bounding box
                box = None
                if T * bres * bres * 4 < hc * wc:
                    box = (int(rects[:, 0].min()), int(rects[:, 1].min()), int(rects[:, 2].max()), int(rects[:, 3].max()))
                self.mask = ops.replay(mine, ops.to_device((areas_yx // df).astype(np.int32)), alpha0, crop_sc,
                                       self.features, self.mask, ops.to_device(off), ops.to_device(lst), **({"box": box} if box else {}))
            elif n_own:
                # pieces in paint order: received strips of earlier foreign tiles (compact [C,h,w] blocks of recv_buf), then
                # my own full tiles (every foreign tile a rank needs precedes its range)
                pieces, prect, o = [], [], 0
                for s_ in range(world):
                    for f, q in recv.get(s_, []):
                        h_, w_ = q[2] - q[0], q[3] - q[1]
                        pieces.append((f, recv_buf[o:o + C * h_ * w_].view(C, h_, w_), q[0], q[1], q[0] - rects[f, 0], q[1] - rects[f, 1]))
                        prect.append(q)
                        o += C * h_ * w_
                order = sorted(range(len(pieces)), key=lambda i: pieces[i][0])
                pieces = [pieces[i][1:] for i in order]
                prect = [prect[i] for i in order]
                for i in range(n_own):
                    pieces.append((mine[i], int(rects[t0 + i, 0]), int(rects[t0 + i, 1]), 0, 0))
                    prect.append(tuple(int(v) for v in rects[t0 + i]))
                off, lst = build_cells(np.asarray(prect, np.int64), hc, wc)
                own_r = rects[t0:t1]
                box = (int(own_r[:, 0].min()), int(own_r[:, 1].min()), int(own_r[:, 2].max()), int(own_r[:, 3].max()))
                self.mask = ops.replay_pieces(pieces, bres, alpha0, crop_sc, self.features, self.mask,
                                              ops.to_device(off), ops.to_device(lst), box)
            if sharded:
                # my canvas now holds the paint sequence under MY tiles only; what it takes to make it whole again
                # (sync_canvas, run lazily by the next sharded call on this canvas) is the tile layout of this call
                self._canvas_rank_local = True
                self._sync_layout = (rects.copy(), bounds)
            # phase 3: last block(s) + ToRGB + compositing on the blended features, own tiles
            for i, (b0, b1) in blist:
                with on_stream(i):
                    outs.append(ops.tail(style(b1 - b0), mine[b0:b1], geom_feats_own[i], pos(b0, b1), bres,
                                         self.render_mode, colors(b1 - b0), sfac, slot=plan_slot(i)))
            join(outs)
        rgba_own = torch.cat(outs) if outs else None
        if not sharded:
            return rgba_own
        buf = torch.zeros([n_pad, R, R, 4], dtype=torch.uint8, device=ops.device)
        if n_own:
            buf[:n_own] = rgba_own
        recv = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        ev = self._comm_event_pair("tile_gather_ms")
        dist.gather(buf, recv, dst=0, group=self.group)
        if ev is not None:
            ev[1].record()
        if rank != 0:
            return None
        return torch.cat([recv[r][:counts[r]] for r in range(world)])

    def _comm_event_pair(self, name: str):
        """A pair of HIP events around a collective of the sharded schedule (first one recorded here, on the caller's stream).
        Only when ``self.comm_timing`` is set (benchmarks: tools/bench_canvas.py): the product path records nothing."""
        dev = getattr(self.ops, "device", None)
        if not getattr(self, "comm_timing", False) or dev is None or getattr(dev, "type", "cpu") != "cuda":
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if not hasattr(self, "_comm_events"):
            self._comm_events = {}
        self._comm_events.setdefault(name, []).append((e0, e1))
        return e0, e1

    def comm_times(self, reset: bool = True) -> Dict[str, float]:
        """Milliseconds this rank's stream spent on the collectives of the sharded calls since the last reset: the part of the
        halo exchange that phase 1 did not hide, and the gather of the RGBA tiles on rank 0 (synchronises the device)."""
        evs = getattr(self, "_comm_events", {})
        if not evs:
            return {}
        torch.cuda.synchronize(self.ops.device)
        out = {k: round(sum(a.elapsed_time(b) for a, b in v), 4) for k, v in evs.items()}
        out["calls"] = max(len(v) for v in evs.values())
        if reset:
            self._comm_events = {}
        return out

    def render_tiles(self, geom_padded: np.ndarray, crops: Sequence[Tuple[int, int]], opts: GanBrushOptions,
                     crop_margin: int = 0, out_canvas: Optional[torch.Tensor] = None):
        """Render the tiles whose top-left corners (y, x) are ``crops`` from the padded geometry [H,W] uint8
        (255 = background) and paste them into ``out_canvas`` [H,W,4] uint8 on the device (rank 0; created if None).
        Equivalent to calling the reference's ``render_stroke`` on the tiles in order with
        ``opts.set_position(x, y)`` (paint_image_main.py:157-177).  Returns the canvas on rank 0, None elsewhere."""
        ops, R = self.ops, self.patch_width
        H, W = geom_padded.shape[:2]
        if self.rows is None:
            self.make_new_canvas(H, W)
        crops = np.asarray([c[:2] for c in crops], np.int64).reshape(-1, 2)
        if crops.shape[0] == 0:
            raise ValueError("no tiles to render")
        df = self.down_factor
        floored = crops if self.feature_blending_level == 0 else (crops // df) * df      # brush.py:253-258
        rgba_all = self._schedule(geom_padded.reshape(H, W), crops, floored, crops, opts, crop_margin)
        if rgba_all is None:
            return None
        if out_canvas is None:
            out_canvas = torch.zeros([H, W, 4], dtype=torch.uint8, device=ops.device)
        m = int(crop_margin)
        off, lst = build_cells(np.concatenate([floored + m, floored + R - m], axis=1), H, W)
        ops.paste(out_canvas, rgba_all, ops.to_device(floored.astype(np.int32)), m, ops.to_device(off), ops.to_device(lst))
        return out_canvas

    graph_strokes = True       # render_stroke replays hipGraph-captured generator passes (TileOps.graph_single)

    def render_stroke(self, stroke_patch: np.ndarray, canvas_patch, opts: GanBrushOptions, meta: Optional[dict] = None):
        """One R x R tile (reference contract, brush.py:244-398): ``stroke_patch`` [R,R,1|4] uint8 with opaque 255 =
        stroke; returns (RGBA uint8 [R-2m, R-2m, 4] numpy, None, {'x','y'}) and updates the feature canvas.
        Single-process call (interactive sessions are one GPU each)."""
        R = self.patch_width
        H, W, _ = stroke_patch.shape
        if W != R or H != R:
            raise RuntimeError("Not implemented")                                  # as the reference, brush.py:273-274
        x = y = m = 0
        if meta is not None:
            x, y = int(meta.get("x")), int(meta.get("y"))
            m = int(meta.get("crop_margin", 0))
        if self.feature_blending_level > 0:
            assert meta is not None, "feature blending needs the tile position"     # brush.py:303
            assert self.rows is not None, "Must call make_new_canvas before rendering with feature blending"
        df = self.down_factor or 1
        fy, fx = (y // df) * df, (x // df) * df
        geom = (255 - stroke_patch[:, :, -1]).astype(np.uint8)                     # back to 255 = background
        pos = None if opts.position is None else opts.position.numpy().reshape(1, 2)
        prev = getattr(self.ops, "graph_single", None)
        if prev is not None and self.graph_strokes:
            self.ops.graph_single = True                  # one tile per call: replay captured generator passes
        try:
            rgba = self._schedule(geom, np.zeros((1, 2), np.int64), np.array([[fy, fx]], np.int64), pos, opts, m)
        finally:
            if prev is not None:
                self.ops.graph_single = prev
        img = rgba[0, m:R - m, m:R - m].cpu().numpy()
        return np.ascontiguousarray(img), None, {"x": fx + m, "y": fy + m}

    def paint_image(self, geom: np.ndarray, opts: GanBrushOptions, crop_margin: int = 10, stitching_mode: str = "all",
                    on_white: bool = False, return_full: bool = False):
        """``paint_image_main.py:145-192`` from the thresholded geometry image [H,W,1] uint8 (255 = background):
        pad, tile with 2*crop_margin overlap, stylize every tile, paste, composite, crop.  Rank 0 returns the image."""
        geom = np.asarray(geom, np.uint8)
        if geom.ndim == 2:
            geom = geom[..., None]
        R = self.patch_width
        padded0 = pad_geo(geom, crop_margin)
        crops, padded = generate_stitching_crops(padded0, R, mode=stitching_mode, overlap_margin=2 * crop_margin)
        self.make_new_canvas(padded.shape[0], padded.shape[1], self.feature_blending_level)
        canvas = self.render_tiles(padded[..., 0], crops, opts, crop_margin=crop_margin)
        if canvas is None:
            return None
        m, (h0, w0) = crop_margin, geom.shape[:2]
        to_host = getattr(self.ops, "to_host", lambda t: t.cpu().numpy())
        result = canvas[m:m + h0, m:m + w0]                          # crop on the device: only the answer crosses PCIe
        if on_white:                                                # paint_image_main.py:179-183 (3 channels out)
            # alpha = a / 255 through a host-computed table: device division is not correctly rounded
            lut = self.ops.to_device(np.arange(256, dtype=np.float32) / np.float32(255))
            a = lut[result[..., 3:].to(torch.int64)]
            result = (result[..., :3].to(torch.float32) * a + 255 * (1 - a)).clip(0, 255).to(torch.uint8)
        out = to_host(result.contiguous())
        if not return_full:
            return out
        full = to_host(canvas)
        return (out, full, crops, padded) if return_full else out
