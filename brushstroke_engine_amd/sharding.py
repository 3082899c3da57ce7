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

from typing import List, Optional, Sequence, Tuple

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


class TileGatherer:
    """Gathers equally shaped per-rank tile tensors to ``dst`` with pre-allocated receive buffers.

    ``start()`` enqueues the gather asynchronously (RCCL runs it on its own stream, so it overlaps the
    next batch's kernels); ``finish()`` waits for it and returns the list of per-rank tensors on ``dst``
    (``None`` elsewhere).  Ragged shards are padded to the largest shard by the caller-provided ``n_max``.
    """

    def __init__(self, shape: Sequence[int], dtype, device, dst: int = 0, group=None):
        self.dst, self.group = dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.recv: Optional[List[torch.Tensor]] = None
        if self.rank == dst and self.world > 1:
            self.recv = [torch.empty(list(shape), dtype=dtype, device=device) for _ in range(self.world)]
        self._work = None
        self._local = None

    def start(self, tiles: torch.Tensor) -> None:
        self._local = tiles
        if self.world == 1:
            return
        self._work = dist.gather(tiles, self.recv if self.rank == self.dst else None, dst=self.dst,
                                 group=self.group, async_op=True)

    def finish(self) -> Optional[List[torch.Tensor]]:
        if self.world == 1:
            return [self._local]
        if self._work is not None:
            self._work.wait()
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
