"""BASELINE config 3's multi-process path on hardware: 2 and 3 ranks (one process each, sharing device 0 of a one-GPU box,
collectives over gloo -- RCCL refuses two ranks on one device) paint ``lamali_sm.png`` and the 9-tile fixture through the
REAL ``TileOps`` (HIP generator, HIP encoder, canvas kernels): the halo strips cross the process group as HIP tensors on
the communication side stream, the pieces replay and the RGBA gather run for real, and rank 0's canvases must equal the
canvases the REFERENCE engine painted exactly as the single-process test demands (tests/test_hip_painting.py).
Launch path = the one ``tools/bench_canvas.py --gpus N`` takes (brushstroke_engine_amd/launch.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, out, mode):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"NB_BENCH_SHARE_GPU": "1", "NB_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "4"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={world}", os.path.join(REPO, "tests", "_canvas_worker.py"), out, mode]
    return subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("world,mode", [(2, "h3"), (3, "f8"), (3, "h3")])
def test_sharded_canvas_real_tileops(tmp_path, world, mode):
    out = str(tmp_path / "canvases.npz")
    r = _launch(world, out, mode)
    assert r.returncode == 0, r.stderr[-4000:]
    res = dict(np.load(out))
    assert int(res["world"]) == world
    lam = load_golden("engine_lamali_r256.npz")
    d = np.abs(res["lamali_level2"].astype(np.int32) - lam["canvas_level2_clear"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (d.max(), (d > 0).mean())
    assert float(res["lamali_mask_sum"]) == lam["feature_canvas_stats"][2]
    d = np.abs(res["lamali_level0"][::37].astype(np.int32) - lam["canvas_level0_rows"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3
    hb = res["lamali_halo_bytes"]
    assert hb[:, 0].sum() == hb[:, 1].sum() > 0
    assert hb.max() < 2 * 64 * 128 * 128 * 4                   # strips, not tiles (one phase-1 tile = 64 ch x 128 x 128 fp32 = 4 MB)
    g = load_golden("engine_r128.npz")
    d = np.abs(res["eng_level2"].astype(np.int32) - g["canvas_level2_clear"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < (5e-3 if mode == "f8" else 1e-3)
    # the second sharded call on the same canvas == the same two calls in ONE process (persistent feature canvas)
    assert bool(res["eng_canvas_equal_on_all_ranks"])
    from brushstroke_engine_amd.networks import Generator
    cfg = cfgmod.style1_config(128)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode=mode).to("cuda")
    ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5)))
    helper = painting.PaintingHelper(ops, batch=2)
    helper.set_feature_blending(2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
    helper.paint_image(g["geom"], opts, crop_margin=int(g["crop_margin"]))
    opts2 = painting.GanBrushOptions()
    opts2.set_style(torch.from_numpy(np.random.RandomState(7).randn(1, cfg.z_dim)), 7)
    ref2 = helper.render_tiles(g["geom_padded"], g["crops"][2:7], opts2, crop_margin=10).cpu().numpy()
    d = np.abs(res["eng_second"].astype(np.int32) - ref2.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    np.testing.assert_allclose(res["eng_features_after_second"], helper.features[0, ::8].cpu().numpy(), atol=1e-5)
    assert np.array_equal(res["eng_mask_after_second"], helper.mask.cpu().numpy())


def test_bench_canvas_self_launches_ranks():
    """`tools/bench_canvas.py --gpus 2` starts its two ranks itself and reports tiles/s, the halo bytes per rank and the
    per-rank phase breakdown; a dying rank is a non-zero exit without a JSON line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"NB_BENCH_SHARE_GPU": "1", "NB_BENCH_BACKEND": "gloo"})
    cmd = [sys.executable, os.path.join(REPO, "tools", "bench_canvas.py"), "--gpus", "2", "--size", "700", "--res", "128", "--level", "2",
           "--batch", "8", "--steps", "2", "--breakdown"]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and len(out["halo_bytes_per_rank"]) == 2
    assert out["halo_bytes_per_rank"][0]["sent"] == out["halo_bytes_per_rank"][1]["received"] > 0
    assert all("head" in b and "tail" in b and "replay_pieces" in b for b in out["breakdown_ms_per_rank"])
    # who took part, and what the two collectives cost each rank's stream
    assert out["rccl"]["world"] == 2 and [r_["rank"] for r_ in out["rccl"]["ranks_seen"]] == [0, 1]
    assert len(out["halo_exchange_ms"]) == 2 and len(out["gather_wait_ms"]) == 2
    assert all(v is not None and v >= 0 for v in out["gather_wait_ms"]) and out["halo_exchange_ms"][1] is not None
    # (the stream probe runs for jobs of >= 6 batches per rank only: this one keeps the default of two streams)
    assert out["n_streams"] in (1, 2) and (out["stream_probe"] is None or out["stream_probe"]["chosen"] == out["n_streams"])
    r = subprocess.run(cmd, env=dict(env, NB_BENCH_FAIL_RANK="1"), cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_lamali_two_ranks_line_names_its_ranks():
    """`tools/bench_lamali.py --gpus 2` (BASELINE config 3 on its named input): the N > 1 line says who took part (`rccl`) and what the
    halo exchange and the tile gather cost each rank, beside the distance from the reference-painted canvas."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"NB_BENCH_SHARE_GPU": "1", "NB_BENCH_BACKEND": "gloo"})
    cmd = [sys.executable, os.path.join(REPO, "tools", "bench_lamali.py"), "--gpus", "2", "--steps", "2"]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl"]["world"] == 2 and out["rccl"]["distinct_devices"] == 1
    assert [r_["rank"] for r_ in out["rccl"]["ranks_seen"]] == [0, 1]
    l2 = out["feature_blending_2"]
    assert l2["vs_reference_canvas"]["max_lsb"] <= 1 and len(l2["halo_exchange_ms"]) == 2 and len(l2["gather_wait_ms"]) == 2
    assert all(v is not None and v >= 0 for v in l2["gather_wait_ms"])
