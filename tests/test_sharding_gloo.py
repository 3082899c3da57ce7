"""N>1 path on CPU: world_size-2 gloo processes shard a tile list, 'render' their shard with a
deterministic stand-in, gather the uint8 RGBA tiles to rank 0 and assemble the canvas."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def fake_render(tile_ids, size):
    """Stand-in for the generator: tile i is filled with a pattern that depends only on i."""
    out = torch.empty([len(tile_ids), size, size, 4], dtype=torch.uint8)
    for j, i in enumerate(tile_ids):
        base = torch.arange(size * size * 4, dtype=torch.int64).reshape(size, size, 4)
        out[j] = ((base * 7 + i * 13) % 251).to(torch.uint8)
    return out


def _worker(rank, world, port, n_tiles, size, tmp):
    sys.path.insert(0, REPO)
    from brushstroke_engine_amd.sharding import TileGatherer, paste_tiles, shard_bounds, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_tiles, rank, world)
        n_max = max(shard_sizes(n_tiles, world))
        mine = fake_render(list(range(lo, hi)), size)
        padded = torch.zeros([n_max, size, size, 4], dtype=torch.uint8)
        padded[: hi - lo] = mine
        g = TileGatherer([n_max, size, size, 4], torch.uint8, "cpu")
        for _ in range(2):                       # two back-to-back steps reuse the receive buffers
            g.start(padded)
            got = g.finish()
        if rank == 0:
            cols = 3
            canvas = torch.zeros([((n_tiles + cols - 1) // cols) * size, cols * size, 4], dtype=torch.uint8)
            for r in range(world):
                a, b = shard_bounds(n_tiles, r, world)
                coords = [((i // cols) * size, (i % cols) * size) for i in range(a, b)]
                paste_tiles(canvas, got[r][: b - a], coords)
            np.save(os.path.join(tmp, "canvas.npy"), canvas.numpy())
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from brushstroke_engine_amd.sharding import shard_bounds, shard_sizes
    for n in (0, 1, 7, 8, 12, 60, 361):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sz = shard_sizes(n, w)
            assert max(sz) - min(sz) <= 1 and sum(sz) == n


@pytest.mark.timeout(120)
def test_two_rank_gather_assembles_canvas(tmp_path):
    n_tiles, size, world = 7, 8, 2           # ragged: 4 + 3 tiles
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_tiles, size, str(tmp_path)), nprocs=world, join=True)
    canvas = np.load(tmp_path / "canvas.npy")
    want = fake_render(list(range(n_tiles)), size).numpy()
    cols = 3
    for i in range(n_tiles):
        y, x = (i // cols) * size, (i % cols) * size
        np.testing.assert_array_equal(canvas[y:y + size, x:x + size], want[i])


# ---------------------------------------------------------------- halo plan (host logic)
def _coverage(plan, rects, bounds, dst):
    """pixel -> set of foreign earlier tiles delivered there, and a flag for duplicate deliveries."""
    got, dup = {}, False
    for (src, d), lst in plan.items():
        if d != dst:
            continue
        for f, (y0, x0, y1, x1) in lst:
            assert bounds[src][0] <= f < bounds[src][1] and src < dst
            ry0, rx0, ry1, rx1 = rects[f]
            assert ry0 <= y0 < y1 <= ry1 and rx0 <= x0 < x1 <= rx1
            for y in range(y0, y1):
                for x in range(x0, x1):
                    s_ = got.setdefault((y, x), set())
                    dup |= f in s_
                    s_.add(f)
    return got, dup


@pytest.mark.parametrize("world,cols,n", [(2, 3, 9), (4, 3, 9), (3, 4, 10), (8, 5, 23), (4, 1, 5)])
def test_halo_plan_delivers_exactly_the_earlier_tiles_under_own_tiles(world, cols, n):
    from brushstroke_engine_amd.sharding import halo_plan, shard_bounds
    hw, stride = 12, 8                                       # overlap 4, as (R/2 - 2*crop) strides do
    rects = np.array([[(i // cols) * stride, (i % cols) * stride, (i // cols) * stride + hw, (i % cols) * stride + hw]
                      for i in range(n)])
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    plan = halo_plan(rects, bounds)
    for dst, (t0, t1) in enumerate(bounds):
        got, dup = _coverage(plan, rects, bounds, dst)
        assert not dup
        want = {}
        for t in range(t0, t1):
            for f in range(t0):
                y0, x0 = max(rects[t][0], rects[f][0]), max(rects[t][1], rects[f][1])
                y1, x1 = min(rects[t][2], rects[f][2]), min(rects[t][3], rects[f][3])
                for y in range(y0, y1):
                    for x in range(x0, x1):
                        want.setdefault((y, x), set()).add(f)
        assert got == want


def test_halo_plan_irregular_tiles():
    """Arbitrary (non-grid) tile positions: strokes painted anywhere on the canvas."""
    from brushstroke_engine_amd.sharding import halo_plan, shard_bounds, rect_subtract
    rs = np.random.RandomState(3)
    yx = rs.randint(0, 40, size=(14, 2))
    rects = np.concatenate([yx, yx + 16], axis=1)
    bounds = [shard_bounds(14, r, 3) for r in range(3)]
    plan = halo_plan(rects, bounds)
    for dst, (t0, t1) in enumerate(bounds):
        got, dup = _coverage(plan, rects, bounds, dst)
        assert not dup
        for (y, x), fs in got.items():
            for f in fs:
                assert any(rects[t][0] <= y < rects[t][2] and rects[t][1] <= x < rects[t][3] for t in range(t0, t1))
    assert rect_subtract((0, 0, 4, 4), (1, 1, 3, 3)) == [(0, 0, 1, 4), (3, 0, 4, 4), (1, 0, 3, 1), (1, 3, 3, 4)]
    assert rect_subtract((0, 0, 4, 4), (5, 5, 6, 6)) == [(0, 0, 4, 4)]
    assert rect_subtract((0, 0, 4, 4), (0, 0, 4, 4)) == []


# ---------------------------------------------------------------- training: gradient all-reduce (BASELINE config 5)
def _grad_worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from brushstroke_engine_amd.training import GanLoss
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        m[2].bias.requires_grad_(False)                                  # a frozen parameter takes no part
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10
        shard = x[rank * 2:(rank + 1) * 2]
        (m(shard) ** 2).mean().backward()
        m[0].bias.grad = None if rank == 1 else m[0].bias.grad          # a rank without a gradient contributes zeros
        n = GanLoss.all_reduce_gradients(m)
        assert n == sum(p.numel() for p in m.parameters() if p.requires_grad)
        if rank == 0:
            torch.save({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, os.path.join(tmp, "g.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_training_gradient_all_reduce_gloo(tmp_path):
    """GanLoss.all_reduce_gradients: the ranks' gradients are averaged with one all-reduce of the flattened gradients --
    equal to the gradient of the mean loss over the whole batch."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "g.pt")
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10
    (0.5 * (m(x[:2]) ** 2).mean() + 0.5 * (m(x[2:]) ** 2).mean()).backward()
    for k, p in m.named_parameters():
        if k == "2.bias":
            assert k not in got
        elif k == "0.bias":                                              # rank 1 contributed zeros for this one
            m2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
            m2.load_state_dict(m.state_dict())
            (m2(x[:2]) ** 2).mean().backward()
            assert torch.allclose(got[k], 0.5 * m2[0].bias.grad, atol=1e-6)
        else:
            assert torch.allclose(got[k], p.grad, atol=1e-6), k
