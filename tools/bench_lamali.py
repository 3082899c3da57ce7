"""BASELINE config 3 on its named input: lamali_sm.png (514 x 800; its thresholded geometry ships in
tests/golden/engine_lamali_r256.npz) through PaintingHelper.paint_image at P = 256, crop margin 10 = 12 tiles, with
feature blending level 2 and 0 -- wall clock from the host geometry array to the host RGBA canvas, and the distance of
the level-2 canvas from the one the REFERENCE engine painted (the fixture).

    python tools/bench_lamali.py [--gpus N] [--steps 10] [--conv-mode f8|h3|f32]

``--gpus N`` launches N ranks itself (see tools/bench_canvas.py, whose measurement this is with --lamali): the 12 tiles are
cut into N contiguous ranges, halo strips exchanged, RGBA tiles gathered on rank 0.  Reference job: neube_stylize.sh:79-85."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_canvas
from brushstroke_engine_amd import launch


def main():
    ap = bench_canvas.parser()
    ap.set_defaults(steps=10, warmup=3)
    a = ap.parse_args()
    a.lamali = True
    if not launch.under_torchrun() and a.gpus > 1:
        raise SystemExit(launch.self_launch(__file__, sys.argv[1:], a.gpus))
    rank, world, dev, backend = launch.init()
    launch.preflight(dev, rank, world)
    fabric = launch.fabric_report(dev, rank, world, backend)
    res = {}
    for level in (2, 0):
        a.level = level
        res[level] = bench_canvas.run(a, rank, world, dev, backend)
    if rank == 0:
        keep = lambda r: {k: r[k] for k in ("value", "unit", "seconds", "tiles", "halo_bytes_per_rank", "vs_reference_canvas", "breakdown_ms",
                                             "breakdown_ms_per_rank", "n_streams", "halo_exchange_ms", "gather_wait_ms", "ms_per_step_per_rank") if k in r}
        line = {"metric": "lamali_sm.png end to end (host geometry -> host RGBA), 12 tiles of 256x256", "n_gpus": world,
                "conv_mode": a.conv_mode, "parallelism": res[2]["parallelism"], "image": res[2]["canvas"],
                "feature_blending_2": keep(res[2]), "feature_blending_0": keep(res[0])}
        if launch.collective(world):
            line["rccl"] = fabric                           # who took part (launch.fabric_report): world, backend, ranks seen, distinct devices
        print(json.dumps(line), flush=True)
    launch.finish(world)


if __name__ == "__main__":
    main()
