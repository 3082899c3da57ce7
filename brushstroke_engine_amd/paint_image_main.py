"""Stylize a large line drawing with a brush style -- the MI355X counterpart of ``forger/viz/paint_image_main.py``
(same flags; ``--gan_checkpoint`` takes the ``.npz`` engine container written by ``tools/convert_snapshot.py`` /
``formats.save_engine_snapshot``).

    python -m brushstroke_engine_amd.paint_image_main --gan_checkpoint engine.npz --geom_image drawing.png \\
        --output_file_prefix out/drawing --style_id 594 --feature_blending_level 2 [--on_white]

``--gpus N`` (or running it under ``torch.distributed.run``, one process per GPU) shards the tiles over the ranks;
rank 0 writes the file.  ``--conv_mode`` chooses the arithmetic of the conv layers explicitly (default: the library
default, ``f8``; ``h3`` = fp32-grade products, ``f32`` = exact fp32 MFMA).
"""
from __future__ import annotations

import argparse
import logging
import os

import numpy as np
import torch
import torch.distributed as dist

from . import encoder, formats, painting
from . import launch
from .networks import DEFAULT_CONV_MODE, Generator

logger = logging.getLogger(__name__)


def set_colors(color_mode: str, brush_options) -> None:
    """``paint_image_main.py:66-85`` for explicit colors: 'r,g,b;r,g,b;r,g,b' (empty entries keep the style's color)."""
    if color_mode in ("1", "2"):
        raise RuntimeError("color modes 1/2 (colors of another style) need StyleUVSMapper.get_colors_raw; pass explicit RGB")
    for i, cspec in enumerate(color_mode.split(";")):
        if cspec:
            rgb = [int(x) for x in cspec.split(",")]
            assert len(rgb) == 3
            brush_options.set_color(i, torch.tensor(rgb, dtype=torch.float32) / 255.0)


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="Tiled canvas stylization on MI355X.")
    ap.add_argument("--gan_checkpoint", required=True, help=".npz engine container (generator + encoder)")
    ap.add_argument("--output_file_prefix", required=True)
    ap.add_argument("--geom_image", required=True, help="Large geometry guidance")
    ap.add_argument("--stitching_mode", default="all", help="Which patches to paint: all | full")
    ap.add_argument("--feature_blending_level", type=int, default=0)
    ap.add_argument("--library", default="rand100")
    ap.add_argument("--style_id", required=True)
    ap.add_argument("--style_id2", default=None)
    ap.add_argument("--style_blend_alpha", type=float, default=0.5)
    ap.add_argument("--crop_margin", type=int, default=10)
    ap.add_argument("--render_mode", default="clear")
    ap.add_argument("--no_uvs_mapping", action="store_true", help="Disable UVS mapping.")
    ap.add_argument("--uvs_calibration", default=None,
                    help=".npz with 'medium' and 'thick' [5,R,R] uint8 calibration drawings (StyleUVSMapper); "
                         "without it UVS mapping is off")
    ap.add_argument("--color_mode", default=None)
    ap.add_argument("--on_white", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--conv_mode", default=None, choices=["f8", "h3", "f32"],
                    help="arithmetic of the conv layers (default: networks.DEFAULT_CONV_MODE = f8: f16 main product + fp8 corrections, "
                         "~1e-4 from fp32 on pixels; h3: three f16 products, 5e-6; f32: exact fp32 MFMA)")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: launch N ranks (one per GPU) and shard the tiles over them")
    return ap


def main(argv=None) -> str:
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and not launch.under_torchrun():
        import sys
        raise SystemExit(launch.self_launch(__file__, sys.argv[1:] if argv is None else list(argv), args.gpus))
    rank, world, device, _ = launch.init()          # kernel library first, then the device and (world > 1) the process group
    cfg, gen_sd, enc_sd, preproc, _ = formats.load_engine_snapshot(args.gan_checkpoint)
    G = Generator(cfg, gen_sd, conv_mode=args.conv_mode or DEFAULT_CONV_MODE).to(device)
    if not encoder.HipGeometryEncoder.supports(cfg.img_resolution):
        raise RuntimeError(f"the geometry-encoder kernels tile patches of 32, 64 or 128 k pixels; this checkpoint paints {cfg.img_resolution}")
    enc = encoder.HipGeometryEncoder(enc_sd, preproc_type=preproc, device=device)
    ops = painting.TileOps(G, enc)
    mapper = None
    if not args.no_uvs_mapping and args.uvs_calibration:
        with np.load(args.uvs_calibration) as z:
            mapper = painting.StyleUVSMapper(ops, z["medium"], z["thick"])
    library = formats.BrushLibrary.from_arg(args.library, z_dim=G.z_dim)
    opts = painting.GanBrushOptions()
    opts.enable_uvs_mapping = mapper is not None
    if args.color_mode is not None:
        set_colors(args.color_mode, opts)
    if args.style_id2 is None:
        library.set_style(args.style_id, opts)
    else:
        library.set_interpolated_style(args.style_id, args.style_id2, args.style_blend_alpha, opts)
    geom = painting.read_geometry_image(args.geom_image)
    helper = painting.PaintingHelper(ops, batch=args.batch, uvs_mapper=mapper)
    helper.set_feature_blending(args.feature_blending_level)
    helper.set_render_mode(args.render_mode)
    with torch.no_grad():
        result = helper.paint_image(geom, opts, crop_margin=args.crop_margin, stitching_mode=args.stitching_mode,
                                    on_white=args.on_white)
    output_file = None
    if result is not None:                                         # rank 0
        style_name = args.style_id
        if args.style_id2 is not None:
            style_name += "_%0.1f%s" % (args.style_blend_alpha, args.style_id2)
        output_file = args.output_file_prefix + "_" + args.render_mode + "_" + str(style_name) + ".png"
        os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
        from PIL import Image
        Image.fromarray(result).save(output_file)
        logger.info(f"Saved result to: {output_file}")
        print(output_file)
    if launch.collective(world):
        dist.barrier()
    return output_file


if __name__ == "__main__":
    main()
