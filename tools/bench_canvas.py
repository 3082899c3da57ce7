"""BASELINE config 3 (tiled canvas stylization) on the HIP path: synthetic line drawing -> tiles -> 3-phase schedule
-> RGBA canvas.  One process per GPU (launch with torch.distributed.run for N>1); prints one JSON line on rank 0.

    python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting  # noqa: E402
from brushstroke_engine_amd.networks import Generator  # noqa: E402


def synthetic_drawing(h, w, seed=0, n_lines=None):
    """[h,w,1] uint8, 255 = background, 0 = stroke: random thick polylines."""
    rs = np.random.RandomState(seed)
    g = np.full((h, w), 255, np.uint8)
    n_lines = n_lines or max(8, (h * w) // 40000)
    for _ in range(n_lines):
        y, x = rs.randint(0, h), rs.randint(0, w)
        ang = rs.rand() * 2 * np.pi
        for _ in range(rs.randint(40, 400)):
            ang += rs.randn() * 0.15
            y, x = y + np.sin(ang) * 2, x + np.cos(ang) * 2
            yi, xi = int(y), int(x)
            if 2 <= yi < h - 2 and 2 <= xi < w - 2:
                g[yi - 2:yi + 3, xi - 2:xi + 3] = 0
    return g[..., None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--crop-margin", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--conv-mode", default="f8")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--encoder", default="hip", choices=["hip"])
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cfg = cfgmod.style1_config(a.res)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode=a.conv_mode).to("cuda")
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5))
    ops = painting.TileOps(G, enc)
    helper = painting.PaintingHelper(ops, batch=a.batch)
    helper.set_feature_blending(a.level)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
    geom = synthetic_drawing(a.size, a.size)
    times = []
    if a.breakdown and world == 1:
        _wrap_timers(ops)
    for i in range(a.warmup + a.steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = helper.paint_image(geom, opts, crop_margin=a.crop_margin)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], device="cuda")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if i >= a.warmup:
            times.append(float(dt))
        elif a.breakdown and world == 1:
            _TIMES.clear()
    if rank == 0:
        n_tiles = len(painting.generate_stitching_crops(painting.pad_geo(geom, a.crop_margin), a.res, 'all', 2 * a.crop_margin)[0])
        t = float(np.mean(times))
        line = {"metric": "tiled canvas stylization, tiles/s (end to end: host tiling + H2D + encoder + generator + "
                          "paste + D2H)", "value": n_tiles / t, "unit": "tiles/s", "n_gpus": world, "seconds": t,
                "tiles": n_tiles, "canvas": [a.size, a.size], "res": a.res, "feature_blending_level": a.level,
                "crop_margin": a.crop_margin, "batch": a.batch, "conv_mode": a.conv_mode, "encoder": a.encoder, "steps": a.steps,
                "stroke_fraction": float((geom == 0).mean())}
        if _TIMES:
            torch.cuda.synchronize()
            line["breakdown_ms"] = {k: round(sum(e0.elapsed_time(e1) for e0, e1 in v) / a.steps, 3) for k, v in _TIMES.items()}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


_TIMES = {}


def _wrap_timers(ops):
    for name in ("geom_tiles", "encode", "head", "tail", "full", "replay", "paste", "map_style"):
        fn = getattr(ops, name)

        def wrapped(*args, _fn=fn, _name=name, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _fn(*args, **kw)
            e1.record()
            _TIMES.setdefault(_name, []).append((e0, e1))
            return r
        setattr(ops, name, wrapped)


if __name__ == "__main__":
    main()
