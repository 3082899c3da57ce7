"""Rank program of tests/test_hip_canvas_sharded.py (started under torchrun by the test; not collected by pytest).
Every rank builds the REAL TileOps (HIP generator + HIP encoder + canvas kernels) on its device and paints, through the
sharded three-phase schedule of PaintingHelper (halo all_to_all_single on a side stream, pieces replay, RGBA gather):
  * lamali_sm.png (R=256, 12 tiles) at feature blending level 2 and 0,
  * the 9-tile fixture (R=128) at level 2, then a SECOND sharded call on the same canvas (lazy canvas sync),
and rank 0 writes the canvases to argv[1] (.npz).  argv[2] = conv mode."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting, launch      # noqa: E402
from brushstroke_engine_amd.networks import Generator                                                          # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    rank, world, dev, backend = launch.init()
    launch.preflight(dev, rank, world)
    from test_painting_cpu import lamali_setup
    from conftest import load_golden
    res = {}
    # --- lamali_sm.png, R = 256 ---
    e = lamali_setup()
    G = Generator(e["cfg"], e["sd"], conv_mode=mode).to(dev)
    ops = painting.TileOps(G, encmod.HipGeometryEncoder(e["esd"], device=dev))
    for level, batch in ((2, 3), (0, 32)):                    # level 2 with ragged batches alternating between the streams
        helper = painting.PaintingHelper(ops, batch=batch)
        helper.set_feature_blending(level)
        opts = painting.GanBrushOptions()
        opts.set_style(torch.from_numpy(e["z"]), 594)
        for rep in range(2):                                    # twice: the second call reuses streams / workspaces / allocator blocks
            r = helper.paint_image(e["geom"], opts, crop_margin=int(e["g"]["crop_margin"]), return_full=True)
        if rank == 0:
            res[f"lamali_level{level}"] = r[1]
        else:
            assert r is None
        if level == 2:
            # (RCCL moves device tensors only; gloo takes either)
            hb = torch.tensor([helper.halo_bytes["sent"], helper.halo_bytes["received"]], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            allhb = [torch.zeros_like(hb) for _ in range(world)]
            dist.all_gather(allhb, hb)
            res["lamali_halo_bytes"] = torch.stack(allhb).cpu().numpy()
            m = helper.mask.clone()
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            res["lamali_mask_sum"] = np.float64(m.sum().item())
    del G, ops
    # --- 9-tile fixture, R = 128: paint, then paint again on the same canvas ---
    g = load_golden("engine_r128.npz")
    cfg = cfgmod.style1_config(128)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode=mode).to(dev)
    ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5), device=dev))
    helper = painting.PaintingHelper(ops, batch=2)
    helper.set_feature_blending(2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
    r = helper.paint_image(g["geom"], opts, crop_margin=int(g["crop_margin"]), return_full=True)
    if rank == 0:
        res["eng_level2"] = r[1]
    opts2 = painting.GanBrushOptions()
    opts2.set_style(torch.from_numpy(np.random.RandomState(7).randn(1, cfg.z_dim)), 7)
    second = helper.render_tiles(g["geom_padded"], g["crops"][2:7], opts2, crop_margin=10)
    helper.sync_canvas()
    torch.cuda.synchronize()
    allf = [torch.empty_like(helper.features) for _ in range(world)]
    dist.all_gather(allf, helper.features)
    same = all(torch.equal(allf[0], f) for f in allf)           # every rank holds the whole canvas afterwards
    if rank == 0:
        res["eng_second"] = second.cpu().numpy()
        res["eng_features_after_second"] = helper.features[0, ::8].cpu().numpy()
        res["eng_mask_after_second"] = helper.mask.cpu().numpy()
        res["eng_canvas_equal_on_all_ranks"] = np.bool_(same)
        res["world"] = np.int64(world)
        np.savez(out_path, **res)
    launch.finish(world)


if __name__ == "__main__":
    main()
