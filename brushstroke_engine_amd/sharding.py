"""Patch-parallel sharding of the generator path over the GPUs of one node.

The reference is strictly single-device (``forger/viz/paint_image_main.py:126``); patches of a batch
(or tiles of a canvas without feature blending) are independent, so they are partitioned over ranks
with replicated weights (~9 MB) and no collective on the data path.  The one real exchange is the
assembly of the stylized result on rank 0: a ``gather`` of the uint8 RGBA tiles
(``[n_rank, R, R, 4]``, 262 KB per 256x256 tile) over RCCL/xGMI -- point-to-point payloads far below
the per-link bandwidth, so a direct gather (not a ring collective) is the right shape.

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first ``n_items % world`` ranks get one extra item."""
    assert 0 <= rank < world and n_items >= 0
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


# ------------------------------------------------------------------------------------------------
# halo plan of the feature-blended canvas
# ------------------------------------------------------------------------------------------------
Rect = Tuple[int, int, int, int]            # (y0, x0, y1, x1), end-exclusive


def rect_subtract(a: Rect, b: Rect) -> List[Rect]:
    """a minus b as up to four disjoint rectangles (top band, bottom band, left and right of the cut)."""
    y0, x0, y1, x1 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    if y0 >= y1 or x0 >= x1:
        return [a]
    out = []
    if a[0] < y0:
        out.append((a[0], a[1], y0, a[3]))
    if y1 < a[2]:
        out.append((y1, a[1], a[2], a[3]))
    if a[1] < x0:
        out.append((y0, a[1], y1, x0))
    if x1 < a[3]:
        out.append((y0, x1, y1, a[3]))
    return out


def halo_plan(rects: np.ndarray, bounds: Sequence[Tuple[int, int]]) -> Dict[Tuple[int, int], List[Tuple[int, Rect]]]:
    """Which strips of which tiles every rank needs from the others to replay the feature-canvas blend of its OWN tiles.

    ``rects`` [T,4] are the tiles' rectangles on the feature canvas in paint order, ``bounds[r] = (t0, t1)`` the
    contiguous ascending tile range rank r owns.  The blended value of tile t at a pixel depends on every EARLIER
    tile covering that pixel (forger/ui/brush.py:190-227 reads what they left on the FeatureCanvas), so rank d needs,
    of every earlier foreign tile f, exactly the pixels f shares with d's tiles.  Returns
    ``{(src, dst): [(f, (y0, x0, y1, x1)), ...]}`` in canvas coordinates, ascending in f; the rectangles of one f
    towards one dst are disjoint (a pixel must meet a tile once in the replay)."""
    r = np.asarray(rects, np.int64).reshape(-1, 4)
    owner = np.empty(r.shape[0], np.int64)
    for k, (a, b) in enumerate(bounds):
        owner[a:b] = k
    plan: Dict[Tuple[int, int], List[Tuple[int, Rect]]] = {}
    for dst, (t0, t1) in enumerate(bounds):
        if t1 <= t0 or t0 == 0:
            continue
        own, ear = r[t0:t1], r[:t0]
        iy0 = np.maximum(ear[:, None, 0], own[None, :, 0]); ix0 = np.maximum(ear[:, None, 1], own[None, :, 1])
        iy1 = np.minimum(ear[:, None, 2], own[None, :, 2]); ix1 = np.minimum(ear[:, None, 3], own[None, :, 3])
        hit = (iy1 > iy0) & (ix1 > ix0)
        for f in np.nonzero(hit.any(axis=1))[0].tolist():
            pieces: List[Rect] = []
            for j in np.nonzero(hit[f])[0].tolist():
                new = [(int(iy0[f, j]), int(ix0[f, j]), int(iy1[f, j]), int(ix1[f, j]))]
                for q in pieces:                                   # keep only what no earlier piece of f covers
                    new = [x for n_ in new for x in rect_subtract(n_, q)]
                pieces.extend(new)
            plan.setdefault((int(owner[f]), dst), []).extend((f, q) for q in pieces)
    return plan


class TileGatherer:
    """Gathers equally shaped per-rank tile tensors to ``dst`` with pre-allocated receive buffers.

    ``start()`` enqueues the gather asynchronously (RCCL runs it on its own stream, so it overlaps the
    next batch's kernels); ``finish()`` waits for it and returns the list of per-rank tensors on ``dst``
    (``None`` elsewhere).  Ragged shards are padded to the largest shard by the caller-provided ``n_max``.
    """

    def __init__(self, shape: Sequence[int], dtype, device, dst: int = 0, group=None, timing: bool = False):
        self.dst, self.group, self.timing = dst, group, bool(timing)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the collective runs with more than one rank -- or at world size 1 under NB_FORCE_PG=1 (launch.py: the N > 1 path through RCCL on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("NB_FORCE_PG") == "1")
        self.recv: Optional[List[torch.Tensor]] = None
        if self.rank == dst and self.active:
            self.recv = [torch.empty(list(shape), dtype=dtype, device=device) for _ in range(self.world)]
        self._work = None
        self._local = None
        # timing=True (benchmarks; off on the product path, where nobody would ever empty the event list):
        # how long finish() kept its caller waiting: host wall clock (a gloo wait blocks the host) and, on a GPU, HIP events on
        # the caller's stream around the wait (an RCCL wait blocks the STREAM, not the host) -- from which a reader of the
        # benchmark line can judge whether the gather was hidden under the next batch
        self.wait_host_ms = 0.0
        self._wait_events = []
        self._cuda = torch.device(device).type == "cuda"
        # (timing events come from a pool created -- and recorded once -- up front: creating HIP events inside a timed region costs a
        #  fresh process milliseconds; wait_ms(reset=True) hands them back)
        self._event_pool = []
        if self.timing and self._cuda and self.active and not torch.cuda.is_current_stream_capturing():
            self._event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 48)]
            for e in self._event_pool:
                e.record()

    def wait_ms(self, reset: bool = True) -> dict:
        """{"host_ms", "stream_ms", "waits"} accumulated by finish() (synchronises the device when events are pending)."""
        stream_ms = None
        if self._wait_events:
            torch.cuda.synchronize()
            stream_ms = round(sum(a.elapsed_time(b) for a, b in self._wait_events), 4)
        out = {"host_ms": round(self.wait_host_ms, 4), "stream_ms": stream_ms, "waits": len(self._wait_events)}
        if reset:
            for a, b in self._wait_events:
                self._event_pool += [a, b]
            self.wait_host_ms, self._wait_events = 0.0, []
        return out

    def start(self, tiles: torch.Tensor) -> None:
        self._local = tiles
        if not self.active:
            return
        self._work = dist.gather(tiles, self.recv if self.rank == self.dst else None, dst=self.dst,
                                 group=self.group, async_op=True)

    def finish(self) -> Optional[List[torch.Tensor]]:
        if not self.active:
            return [self._local]
        if self._work is not None and not self.timing:
            self._work.wait()
            self._work = None
        elif self._work is not None:
            import time as _time
            ev = None
            if self._cuda and not torch.cuda.is_current_stream_capturing():
                ev = ((self._event_pool.pop(), self._event_pool.pop()) if len(self._event_pool) >= 2 else
                      (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                ev[0].record()
            t0 = _time.perf_counter()
            self._work.wait()
            self.wait_host_ms += (_time.perf_counter() - t0) * 1e3
            if ev is not None:
                ev[1].record()
                self._wait_events.append(ev)
            self._work = None
        return self.recv if self.rank == self.dst else None


def paste_tiles(canvas: torch.Tensor, tiles: torch.Tensor, coords: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """Write tiles [n,h,w,4] into an HWC canvas at (y,x) = coords[i], clipping at the border
    (``forger/viz/paint_image_main.py:173-177``)."""
    H, W = canvas.shape[:2]
    for t, (y, x) in zip(tiles, coords):
        h = min(t.shape[0], H - y)
        w = min(t.shape[1], W - x)
        if h > 0 and w > 0:
            canvas[y:y + h, x:x + w] = t[:h, :w]
    return canvas
