"""BASELINE config 3 (tiled canvas stylization) on the HIP path: synthetic line drawing (or, with --lamali, the job's
named input lamali_sm.png from its fixture) -> tiles -> 3-phase schedule -> RGBA canvas on rank 0's host.
One process per GPU; prints one JSON line on rank 0.

    python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 [--gpus N] [--breakdown] [--lamali]

``--gpus N`` (N > 1) without a torchrun environment launches the N ranks itself (child torchrun before anything here
touches the GPU, brushstroke_engine_amd/launch.py) and exits with the child's code: a rank that dies or a failed
collective pre-flight is a non-zero exit without a JSON line.  With N > 1 the tiles are cut into contiguous per-rank
ranges; the line reports tiles/s (max-over-ranks wall clock between barriers), the halo bytes every rank sent / received
and, with --breakdown, each rank's per-phase device times.  Reference job: neube_stylize.sh:79-85 +
forger/viz/paint_image_main.py:145-192 (single device there).
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
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting, launch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
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


def lamali_geometry():
    g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "engine_lamali_r256.npz")))
    h, w = g["geom_shape"].tolist()
    return (np.unpackbits(g["geom_bits"])[:h * w].reshape(h, w) * 255).astype(np.uint8)[..., None], g


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--crop-margin", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--conv-mode", default="f8")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--lamali", action="store_true", help="paint lamali_sm.png (tests/golden/engine_lamali_r256.npz, R=256) instead of the "
                                                            "synthetic drawing and report the distance to the reference-painted canvas")
    ap.add_argument("--encoder", default="hip", choices=["hip"])
    ap.add_argument("--streams", default="auto", choices=["auto", "1", "2", "3", "ab"],
                    help="batch streams of the tiled schedule: auto = TileOps.choose_streams' probe (the default of the library), "
                         "1 / 2 / 3 = fixed, ab = time the job with 1, with 2 and with the probe's choice on this box and report all")
    return ap


def main():
    a = parser().parse_args()
    if not launch.under_torchrun() and a.gpus > 1:
        raise SystemExit(launch.self_launch(__file__, sys.argv[1:], a.gpus))
    rank, world, dev, backend = launch.init()
    launch.preflight(dev, rank, world)
    fabric = launch.fabric_report(dev, rank, world, backend)
    if a.streams == "ab":
        # the same job with one batch stream, with two, and with what the probe picks -- one box, one process, back to back
        ab = {}
        for pol in ("1", "2", "auto"):
            a.streams = pol
            l_ = run(a, rank, world, dev, backend)
            if rank == 0:
                ab[pol] = {"seconds": l_["seconds"], "tiles_per_s": l_["value"], "stream_probe": l_.get("stream_probe"), "n_streams": l_.get("n_streams")}
        line = l_
        if rank == 0:
            line["stream_ab"] = ab
    else:
        line = run(a, rank, world, dev, backend)
    if rank == 0:
        if launch.collective(world):
            line["rccl"] = fabric
        print(json.dumps(line), flush=True)
    launch.finish(world)


def run(a, rank, world, dev, backend):
    """One configuration on an initialised process (group); returns the JSON-able result on rank 0, None elsewhere."""
    _TIMES.clear()
    multi = launch.collective(world)             # N > 1, or NB_FORCE_PG=1 at N = 1 (the same branches through RCCL on one GPU)
    gold = None
    if a.lamali:
        a.res = 256
        geom, gold = lamali_geometry()
    else:
        geom = synthetic_drawing(a.size, a.size)
    cfg = cfgmod.style1_config(a.res)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=int(gold["weights_seed"]) if gold else 0), conv_mode=a.conv_mode).to(dev)
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(int(gold["encoder_seed"]) if gold else 5), device=dev)
    ops = painting.TileOps(G, enc)
    ops.stream_policy = 0 if a.streams == "auto" else int(a.streams)
    helper = painting.PaintingHelper(ops, batch=a.batch)
    helper.comm_timing = True                            # (HIP events around the collectives: benchmark only)
    helper.set_feature_blending(a.level)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(np.random.RandomState(int(gold["style_seed"]) if gold else 594).randn(1, cfg.z_dim)), 594)
    times = []
    if a.breakdown:
        _wrap_timers(ops)
    full = line = None
    for i in range(a.warmup + a.steps):
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if gold is not None and i == a.warmup + a.steps - 1:
            res = helper.paint_image(geom, opts, crop_margin=a.crop_margin, return_full=True)
            full = None if res is None else res[1]
        else:
            helper.paint_image(geom, opts, crop_margin=a.crop_margin)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], device=dev)
        if multi:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if i >= a.warmup:
            times.append(float(dt))
        else:
            helper.comm_times()                          # (reset: collectives of the timed steps only)
            if a.breakdown:
                _TIMES.clear()
    # per-rank figures: halo bytes of the exchange, per-phase device time
    torch.cuda.synchronize()
    comm = helper.comm_times() if multi else {}
    mine = {"rank": rank, "halo_bytes": helper.halo_bytes, "seconds": float(np.mean(times)) if times else None,
            "comm_ms_per_step": {k: round(v / max(1, a.steps), 4) for k, v in comm.items() if k != "calls"},
            "breakdown_ms": {k: round(sum(e0.elapsed_time(e1) for e0, e1 in v) / a.steps, 3) for k, v in _TIMES.items()} if _TIMES else None}
    per_rank = [mine]
    if multi:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        n_tiles = len(painting.generate_stitching_crops(painting.pad_geo(geom, a.crop_margin), a.res, 'all', 2 * a.crop_margin)[0])
        t = float(np.mean(times))
        line = {"metric": "tiled canvas stylization, tiles/s (end to end: host tiling + H2D + encoder + generator + "
                          "paste + D2H)", "value": n_tiles / t, "unit": "tiles/s", "n_gpus": world, "seconds": t,
                "tiles": n_tiles, "canvas": list(geom.shape[:2]), "input": "lamali_sm.png (fixture)" if gold else "synthetic line drawing",
                "res": a.res, "feature_blending_level": a.level,
                "crop_margin": a.crop_margin, "batch": a.batch, "conv_mode": a.conv_mode, "encoder": a.encoder, "steps": a.steps,
                "stroke_fraction": float((geom == 0).mean()),
                "parallelism": "single GPU" if not multi else
                               f"tiles in {world} contiguous ranges; halo strips by one all_to_all_single ({'RCCL' if backend == 'nccl' else backend}) "
                               f"under phase 1, RGBA tiles gathered on rank 0",
                "timing": "max over ranks of the wall clock between barriers, mean over steps"}
        line["n_streams"] = getattr(ops, "n_streams", None)
        line["stream_probe"] = getattr(ops, "stream_probe", None)
        if multi:
            line["halo_bytes_per_rank"] = [p["halo_bytes"] for p in per_rank]
            # what the collectives cost each rank's stream per painted canvas: the part of the halo exchange phase 1 did not
            # hide, and the gather of the RGBA tiles (rank 0 waits for everybody's); HIP events around the waits
            line["halo_exchange_ms"] = [p["comm_ms_per_step"].get("halo_exchange_exposed_ms") for p in per_rank]
            line["gather_wait_ms"] = [p["comm_ms_per_step"].get("tile_gather_ms") for p in per_rank]
        if a.breakdown:
            line["breakdown_ms" if world == 1 else "breakdown_ms_per_rank"] = per_rank[0]["breakdown_ms"] if world == 1 else [p["breakdown_ms"] for p in per_rank]
        if gold is not None and a.level in (0, 2) and f"canvas_level{a.level}_clear" in gold and full is not None:
            d = np.abs(full.astype(np.int32) - gold[f"canvas_level{a.level}_clear"].astype(np.int32))
            line["vs_reference_canvas"] = {"max_lsb": int(d.max()), "bytes_differing": float((d > 0).mean())}
        return line
    return None


_TIMES = {}


def _wrap_timers(ops):
    for name in ("geom_tiles", "encode", "head", "tail", "full", "replay", "replay_pieces", "paste", "map_style"):
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
