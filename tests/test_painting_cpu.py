"""CPU tests of the painting-engine rows (SURVEY 8 e/f1/f2): the oracle against the reference-generated canvases,
the host logic of brushstroke_engine_amd.painting, and the three-phase (sharded) schedule with an oracle-backed
device stand-in -- world size 1 and world size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, REPO
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
from oracle import neube_oracle as no, painting_oracle as po
from oracle_tile_ops import OracleTileOps, sequential_replay


@pytest.fixture(scope="module")
def eng():
    g = load_golden("engine_r128.npz")
    cfg = cfgmod.style1_config(int(g["resolution"]))
    sd = wmod.random_state_dict(cfg, seed=int(g["weights_seed"]))
    esd = encmod.random_encoder_state_dict(int(g["encoder_seed"]))
    z = np.random.RandomState(int(g["style_seed"])).randn(1, cfg.z_dim)
    return dict(g=g, cfg=cfg, sd=sd, esd=esd, z=z)


@pytest.fixture(scope="module")
def lam():
    """BASELINE config 3 on its named input: lamali_sm.png (514 x 800) painted by the REFERENCE engine at P = 256, crop
    margin 10, blending level 2 -- 12 tiles (tests/golden/make_golden_engine.py --lamali)."""
    return lamali_setup()


def lamali_setup():
    g = load_golden("engine_lamali_r256.npz")
    h, w = g["geom_shape"].tolist()
    geom = (np.unpackbits(g["geom_bits"])[:h * w].reshape(h, w) * 255).astype(np.uint8)
    cfg = cfgmod.style1_config(int(g["resolution"]))
    sd = wmod.random_state_dict(cfg, seed=int(g["weights_seed"]))
    esd = encmod.random_encoder_state_dict(int(g["encoder_seed"]))
    z = np.random.RandomState(int(g["style_seed"])).randn(1, cfg.z_dim)
    return dict(g=g, geom=geom, cfg=cfg, sd=sd, esd=esd, z=z)


def _canvas_close(a, b, max_frac=2e-4):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1, d.max()                    # fp32 reassociation can flip a uint8 truncation
    assert (d > 0).mean() <= max_frac, (d > 0).sum()


# ---------------------------------------------------------------- oracle pinned by the reference
def test_oracle_encoder_matches_reference(eng):
    g = eng["g"]
    f = po.encoder_encode(eng["esd"], torch.from_numpy(g["enc_in"]))
    np.testing.assert_allclose(f[0].numpy(), g["enc_f0"], atol=1e-6)
    np.testing.assert_allclose(f[1].numpy()[:, ::8], g["enc_f1"], atol=1e-6)


@pytest.mark.parametrize("level", [0, 2])
def test_oracle_painter_matches_reference_canvas(eng, level):
    g = eng["g"]
    P = po.OraclePainter(no.OracleGenerator(eng["cfg"], eng["sd"]), eng["esd"])
    out, full, crops, padded = P.paint_image(g["geom"][..., None], z=eng["z"], crop_margin=int(g["crop_margin"]),
                                             feature_blending=level)
    assert np.array_equal(np.array([c[:2] for c in crops]), g["crops"])
    assert np.array_equal(padded[..., 0], g["geom_padded"])
    _canvas_close(full, g[f"canvas_level{level}_clear"])
    assert out.shape == (g["geom"].shape[0], g["geom"].shape[1], 4)
    if level == 2:
        st = g["feature_canvas_stats"]
        assert float(P.mask.sum()) == st[2]
        np.testing.assert_allclose(float(P.features.double().sum()), st[0], rtol=1e-6)
        np.testing.assert_allclose(P.features[0, ::16, ::4, ::4].numpy(), g["feature_canvas_sub"], atol=2e-5)


def test_oracle_painter_matches_reference_lamali(lam):
    """Config 3's named workload through the oracle's sequential tile loop."""
    g = lam["g"]
    P = po.OraclePainter(no.OracleGenerator(lam["cfg"], lam["sd"]), lam["esd"])
    out, full, crops, padded = P.paint_image(lam["geom"][..., None], z=lam["z"], crop_margin=int(g["crop_margin"]),
                                             feature_blending=2)
    assert len(crops) == 12 and np.array_equal(np.array([c[:2] for c in crops]), g["crops"])
    _canvas_close(full, g["canvas_level2_clear"])
    st = g["feature_canvas_stats"]
    assert float(P.mask.sum()) == st[2]
    np.testing.assert_allclose(float(P.features.double().sum()), st[0], rtol=1e-5)
    np.testing.assert_allclose(P.features[0, ::16, ::8, ::8].numpy(), g["feature_canvas_sub"], atol=5e-5)
    assert out.shape == (800, 514, 4)


# ---------------------------------------------------------------- host logic
@pytest.mark.parametrize("width,margin,crop", [(64, 8, 5), (128, 16, 10), (128, 8, 0), (32, 4, 2), (64, 8, 0)])
def test_dirty_area_alpha_bitwise(width, margin, crop):
    a = painting.dirty_area_alpha(width, margin, crop)
    b = po.dirty_area_alpha(width, margin, crop).numpy()
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert a.max() == 1 and a.min() == 0 and a[width // 2, width // 2] == 1


def test_tiling_matches_reference(eng):
    g = eng["g"]
    m = int(g["crop_margin"])
    crops, padded = painting.generate_stitching_crops(painting.pad_geo(g["geom"][..., None], m), 128, "all", 2 * m)
    assert np.array_equal(np.array([c[:2] for c in crops]), g["crops"])
    assert np.array_equal(padded[..., 0], g["geom_padded"])
    full_only, _ = painting.generate_stitching_crops(painting.pad_geo(g["geom"][..., None], m), 128, "full", 2 * m)
    assert 0 < len(full_only) <= len(crops)
    empty = np.full((40, 50, 1), 255, np.uint8)                       # ragged: image smaller than one patch
    c2, p2 = painting.generate_stitching_crops(empty, 128, "all", 20)
    assert c2 == [(0, 0, 128, 128)] and p2.shape == (88 + 128, 88 + 128, 1)
    assert painting.generate_stitching_crops(empty, 128, "full", 20)[0] == []


def test_build_cells_bruteforce():
    rs = np.random.RandomState(0)
    h, w = 37, 150
    rects = []
    for _ in range(12):
        y0, x0 = rs.randint(-5, h), rs.randint(-5, w)
        rects.append((y0, x0, y0 + rs.randint(1, 30), x0 + rs.randint(1, 90)))
    rects.append((0, 0, 0, 0))                                         # padding row
    off, lst = painting.build_cells(np.array(rects), h, w)
    ncx = -(-w // painting.CELL_W)
    for cy in range(-(-h // painting.CELL_H)):
        for cx in range(ncx):
            want = [t for t, (y0, x0, y1, x1) in enumerate(rects)
                    if max(y0, cy * painting.CELL_H, 0) < min(y1, (cy + 1) * painting.CELL_H, h)
                    and max(x0, cx * painting.CELL_W, 0) < min(x1, (cx + 1) * painting.CELL_W, w)]
            c = cy * ncx + cx
            assert list(lst[off[c]:off[c + 1]]) == want


def test_otsu_and_geometry_preparation():
    rs = np.random.RandomState(1)
    img = np.where(rs.rand(64, 64) < 0.3, rs.randint(10, 40, (64, 64)), rs.randint(180, 250, (64, 64))).astype(np.uint8)
    t = painting.threshold_otsu(img)
    assert 39 <= t < 180                                               # any cut between the two modes is optimal
    # exhaustive between-class variance maximisation
    best = max(range(int(img.min()), int(img.max())),
               key=lambda k: (img <= k).sum() * (img > k).sum() * (img[img <= k].mean() - img[img > k].mean()) ** 2)
    assert t == best
    rgba = np.zeros((32, 32, 4), np.uint8)
    rgba[8:12, :, 3] = 255                                             # an opaque black line on transparent
    g = painting.prepare_geometry_image(rgba)
    assert g.shape == (32, 32, 1) and set(np.unique(g)) == {0, 255} and (g[8:12] == 0).all() and (g[:8] == 255).all()
    assert (painting.prepare_geometry_image(255 - rgba[..., 3]) == g).all()      # gray input, same drawing


def test_encoder_state_table_and_errors(eng):
    """The encoder's key / shape table (what random_encoder_state_dict fills and strong.pt-style checkpoints carry) and the
    loud failures: no GPU -> no encoder (there is no torch / CPU module of it in the package)."""
    sd = encmod.random_encoder_state_dict(5)
    assert [k for k, _ in encmod.ENCODER_STATE_SHAPES] == list(sd.keys()) and len(sd) == 65
    assert sd["decoder.model.0.conv.conv.0.weight"].shape == (256, 16, 3, 3) and sd["encoder.model.0.conv.0.weight"].shape == (64, 1, 7, 7)
    assert not hasattr(encmod, "GeometryEncoder") and not hasattr(encmod, "build_encoder")
    with pytest.raises(RuntimeError):
        encmod.HipGeometryEncoder(sd, device="cpu")
    with pytest.raises(RuntimeError):
        encmod.HipGeometryEncoder(sd, preproc_type="bogus", device="cpu")
    assert all(encmod.HipGeometryEncoder.supports(r) for r in (32, 64, 128, 256, 512)) and not any(encmod.HipGeometryEncoder.supports(r) for r in (16, 48, 96, 192))


def test_tile_ops_needs_gpu(eng):
    class FakeG:
        img_resolution = 128
        cfg = eng["cfg"]
    with pytest.raises(RuntimeError):
        painting.TileOps(FakeG(), None, device="cpu")


# ---------------------------------------------------------------- the schedule on CPU (oracle stand-in)
def _paint(eng, level, batch=4, group=None):
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    helper = painting.PaintingHelper(ops, batch=batch, group=group)
    helper.set_feature_blending(level)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    res = helper.paint_image(eng["g"]["geom"], opts, crop_margin=int(eng["g"]["crop_margin"]), return_full=True)
    return helper, res


@pytest.mark.parametrize("level", [0, 2])
def test_three_phase_schedule_matches_reference_canvas(eng, level):
    helper, (out, full, crops, padded) = _paint(eng, level)
    _canvas_close(full, eng["g"][f"canvas_level{level}_clear"])
    if level == 2:
        st = eng["g"]["feature_canvas_stats"]
        assert float(helper.mask.sum()) == st[2]
        np.testing.assert_allclose(helper.features[0, ::16, ::4, ::4].numpy(), eng["g"]["feature_canvas_sub"], atol=2e-5)


def test_render_stroke_sequence_equals_batched_schedule(eng):
    """The reference contract (one render_stroke per tile, canvas state in between) == one render_tiles call."""
    g = eng["g"]
    m = int(g["crop_margin"])
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    helper = painting.PaintingHelper(ops)
    padded = g["geom_padded"]
    helper.make_new_canvas(padded.shape[0], padded.shape[1], feature_blending=2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    result = np.zeros(padded.shape + (4,), np.uint8)
    for y, x in g["crops"][:5].tolist():
        opts.set_position(x, y)
        patch = (255 - padded[y:y + 128, x:x + 128])[..., None]
        res, _, meta = helper.render_stroke(patch, None, opts, meta={"x": x, "y": y, "crop_margin": m})
        assert res.shape == (128 - 2 * m, 128 - 2 * m, 4) and meta == {"x": x + m, "y": y + m}
        result[meta["y"]:meta["y"] + res.shape[0], meta["x"]:meta["x"] + res.shape[1]] = res
    h2 = painting.PaintingHelper(ops)
    h2.make_new_canvas(padded.shape[0], padded.shape[1], feature_blending=2)
    batched = h2.render_tiles(padded, g["crops"][:5], opts, crop_margin=m).numpy()
    _canvas_close(result, batched)
    assert torch.equal(helper.mask, h2.mask)
    np.testing.assert_allclose(helper.features.numpy(), h2.features.numpy(), atol=1e-5)
    with pytest.raises(RuntimeError):
        helper.render_stroke(np.zeros((64, 64, 1), np.uint8), None, opts)
    with pytest.raises(RuntimeError):
        helper.set_render_mode("bogus")


def test_user_colors_and_full_mode(eng):
    g = eng["g"]
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    helper = painting.PaintingHelper(ops)
    helper.set_render_mode("full")
    opts = painting.GanBrushOptions(primary_color=np.array([255, 0, 0], np.uint8))
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    padded = g["geom_padded"]
    out = helper.render_tiles(padded, g["crops"][:2], opts, crop_margin=0).numpy()
    assert (out[:128, :128, 3] == 255).all()
    P = po.OraclePainter(no.OracleGenerator(eng["cfg"], eng["sd"]), eng["esd"])
    P.render_mode = "full"
    y, x = g["crops"][1].tolist()
    ref, _ = P.render_stroke((255 - padded[y:y + 128, x:x + 128])[..., None], z=eng["z"], x=x, y=y, position=(y, x),
                             user_colors=opts.user_colors())
    _canvas_close(out[y:y + 128, x:x + 128], ref)


@pytest.mark.parametrize("res", [64, 32])
def test_oracle_encoder_small_patch_sizes(res):
    """The encoder restatement against the REFERENCE encoder at patch sizes 64 and 32 (tests/golden/encoder_small.npz)."""
    g = load_golden("encoder_small.npz")
    f = po.encoder_encode(encmod.random_encoder_state_dict(int(g["encoder_seed"])), torch.from_numpy(g[f"enc_in_r{res}"]))
    np.testing.assert_allclose(f[0].numpy(), g[f"enc_f0_r{res}"], atol=2e-6)
    np.testing.assert_allclose(f[1].numpy(), g[f"enc_f1_r{res}"], atol=2e-6)


def test_oracle_twenty_stroke_session(eng):
    """The painting oracle against the REFERENCE's 20-stroke interactive session (three alternating styles, one canvas,
    feature blending 2; tests/golden/make_golden_engine.py --strokes): final canvas and the feature canvas after 5 / 10 / 20
    strokes.  Pins the oracle for state carried across strokes with changing styles."""
    g = load_golden("engine_strokes_r128.npz")
    R = int(g["resolution"])
    patches = (np.unpackbits(g["patches"])[:20 * R * R].reshape(20, R, R, 1) * 255).astype(np.uint8)
    m, size = int(g["crop_margin"]), int(g["size"])
    P = po.OraclePainter(no.OracleGenerator(eng["cfg"], eng["sd"]), eng["esd"])
    P.make_new_canvas(size, size, feature_blending=2)
    result = np.zeros((size, size, 4), np.uint8)
    for i in range(20):
        x, y = g["xy"][i].tolist()
        z = np.random.RandomState(int(g["styles"][i])).randn(1, eng["cfg"].z_dim)
        res, meta = P.render_stroke(patches[i], z=z, x=x, y=y, crop_margin=m, position=(y, x))
        result[meta["y"]:meta["y"] + res.shape[0], meta["x"]:meta["x"] + res.shape[1]] = res
        if i + 1 in (5, 10, 20):
            np.testing.assert_allclose(P.features[0, ::8, ::4, ::4].numpy(), g[f"feature_canvas_sub_{i + 1}"], atol=2e-5)
            assert float(P.mask.sum()) == float(g[f"feature_canvas_mask_sum_{i + 1}"])
    _canvas_close(result, g["canvas"], max_frac=5e-4)


# ---------------------------------------------------------------- world size 2 (gloo)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.set_num_threads(4)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("engine_r128.npz")
        cfg = cfgmod.style1_config(128)
        e = dict(g=g, cfg=cfg, sd=wmod.random_state_dict(cfg, seed=0), esd=encmod.random_encoder_state_dict(5),
                 z=np.random.RandomState(594).randn(1, cfg.z_dim), pad2=g["geom_padded"])
        for level in (0, 2):
            helper, res = _paint(e, level, batch=2)
            if rank == 0:
                np.save(os.path.join(tmp, f"full{level}.npy"), res[1])
            else:
                assert res is None
            if level == 2:
                # the feature canvas is rank-local after the halo schedule: a rank's mask covers what was painted
                # under ITS tiles; their union is the reference's mask
                m = helper.mask.clone()
                dist.all_reduce(m, op=dist.ReduceOp.MAX)
                hb = torch.tensor([helper.halo_bytes["sent"], helper.halo_bytes["received"]], dtype=torch.int64)
                allhb = [torch.zeros_like(hb) for _ in range(world)]
                dist.all_gather(allhb, hb)
                if rank == 0:
                    np.save(os.path.join(tmp, "mask.npy"), m.numpy())
                    np.save(os.path.join(tmp, "halo_bytes.npy"), torch.stack(allhb).numpy())
                # painting on the SAME canvas again across ranks (the reference keeps one persistent FeatureCanvas over
                # strokes, brush.py:33-92): the second sharded call first makes the canvas whole on every rank
                # (sync_canvas: one all-reduce in which every pixel has exactly one owner), then blends against it
                opts2 = painting.GanBrushOptions()
                opts2.set_style(torch.from_numpy(np.random.RandomState(7).randn(1, cfg.z_dim)), 7)
                second = helper.render_tiles(e["pad2"], g["crops"][2:7], opts2, crop_margin=10)
                helper.sync_canvas()
                if rank == 0:
                    np.save(os.path.join(tmp, "second.npy"), second.numpy())
                np.save(os.path.join(tmp, f"features_after_second_{rank}.npy"), helper.features.numpy())
                np.save(os.path.join(tmp, f"mask_after_second_{rank}.npy"), helper.mask.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_lamali(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.set_num_threads(max(1, 8 // world))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        e = lamali_setup()
        ops = OracleTileOps(e["cfg"], e["sd"], e["esd"])
        helper = painting.PaintingHelper(ops, batch=2)
        helper.set_feature_blending(2)
        opts = painting.GanBrushOptions()
        opts.set_style(torch.from_numpy(e["z"]), 594)
        res = helper.paint_image(e["geom"], opts, crop_margin=int(e["g"]["crop_margin"]), return_full=True)
        if rank == 0:
            np.save(os.path.join(tmp, "lamali_full.npy"), res[1])
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_lamali_sharded_world3_gloo(lam, tmp_path):
    """Config 3: lamali_sm.png, 12 tiles (4 rows x 3) over 3 ranks (4 + 4 + 4: every range ends mid-row), halo
    exchange + RGBA gather, against the canvas the reference engine painted."""
    mp.spawn(_worker_lamali, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    _canvas_close(np.load(tmp_path / "lamali_full.npy"), lam["g"]["canvas_level2_clear"])


def _worker_one_tile(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.set_num_threads(4)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("engine_r128.npz")
        cfg = cfgmod.style1_config(128)
        ops = OracleTileOps(cfg, wmod.random_state_dict(cfg, seed=0), encmod.random_encoder_state_dict(5))
        helper = painting.PaintingHelper(ops, batch=2)
        helper.make_new_canvas(g["geom_padded"].shape[0], g["geom_padded"].shape[1], feature_blending=2)
        opts = painting.GanBrushOptions()
        opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
        out = helper.render_tiles(g["geom_padded"], g["crops"][:1], opts, crop_margin=10)      # 1 tile, 2 ranks
        if rank == 0:
            np.save(os.path.join(tmp, "one.npy"), out.numpy())
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_fewer_tiles_than_ranks_gloo(eng, tmp_path):
    """A rank without tiles still takes part in the all_gather / gather (padded, empty shard)."""
    mp.spawn(_worker_one_tile, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    helper = painting.PaintingHelper(ops)
    helper.make_new_canvas(eng["g"]["geom_padded"].shape[0], eng["g"]["geom_padded"].shape[1], feature_blending=2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    ref = helper.render_tiles(eng["g"]["geom_padded"], eng["g"]["crops"][:1], opts, crop_margin=10).numpy()
    assert np.array_equal(np.load(tmp_path / "one.npy"), ref)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_schedule_gloo(eng, tmp_path, world):
    """9 tiles (3 x 3) over 2 ranks (5 + 4) and over 4 ranks (3 + 2 + 2 + 2: the ranges end mid-row, so right, down and
    both diagonal neighbours cross ranks): halo exchange of the overlapped strips only, padded RGBA gather."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for level in (0, 2):
        _canvas_close(np.load(tmp_path / f"full{level}.npy"), eng["g"][f"canvas_level{level}_clear"])
    assert float(np.load(tmp_path / "mask.npy").sum()) == eng["g"]["feature_canvas_stats"][2]
    # exchanged volume: strips only -- far below one whole phase-1 tile (128 ch x 64 x 64 fp32 = 2 MB) per boundary tile
    hb = np.load(tmp_path / "halo_bytes.npy")
    assert hb[:, 0].sum() == hb[:, 1].sum() > 0
    tile_bytes = 128 * 64 * 64 * 4
    assert hb.max() < 2 * tile_bytes, hb                      # (the old schedule moved 9 whole tiles to every rank)
    # a second sharded call on the same canvas == the same two calls in one process (one persistent feature canvas)
    helper, res = _paint(eng, 2, batch=2)
    opts2 = painting.GanBrushOptions()
    opts2.set_style(torch.from_numpy(np.random.RandomState(7).randn(1, eng["cfg"].z_dim)), 7)
    ref2 = helper.render_tiles(eng["g"]["geom_padded"], eng["g"]["crops"][2:7], opts2, crop_margin=10).numpy()
    _canvas_close(np.load(tmp_path / "second.npy"), ref2)
    for r in range(world):                                    # ... and every rank holds the whole canvas afterwards
        np.testing.assert_allclose(np.load(tmp_path / f"features_after_second_{r}.npy"), helper.features.numpy(), atol=2e-5)
        assert np.array_equal(np.load(tmp_path / f"mask_after_second_{r}.npy"), helper.mask.numpy())


# ---------------------------------------------------------------- clear-background (UVS) mapping
def test_oracle_uvs_mapping_matches_reference(eng):
    g = eng["g"]
    out = no.map_style_s(torch.tensor(np.float32(1.7)), torch.from_numpy(g["uvsmap_in"]))
    np.testing.assert_array_equal(out.numpy(), g["uvsmap_out"])
    assert (out.sum(dim=1) - 1).abs().max() < 1e-5 or (out[:, 2] == 1).any()
    P = po.OraclePainter(no.OracleGenerator(eng["cfg"], eng["sd"]), eng["esd"])
    sf = P.compute_sfactor(g["uvs_cal_medium"], g["uvs_cal_thick"], z=eng["z"])
    np.testing.assert_allclose(float(sf), float(g["uvs_sfactor"]), rtol=1e-5)
    P.sfactor = sf
    _, full, _, _ = P.paint_image(g["geom"][..., None], z=eng["z"], crop_margin=int(g["crop_margin"]), feature_blending=2)
    _canvas_close(full, g["canvas_level2_clear_uvsmap"])


def test_schedule_with_uvs_mapping(eng):
    g = eng["g"]
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    mapper = painting.StyleUVSMapper(ops, g["uvs_cal_medium"], g["uvs_cal_thick"])
    helper = painting.PaintingHelper(ops, batch=4, uvs_mapper=mapper)
    helper.set_feature_blending(2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    opts.enable_uvs_mapping = True
    _, full, _, _ = helper.paint_image(g["geom"], opts, crop_margin=int(g["crop_margin"]), return_full=True)
    np.testing.assert_allclose(float(mapper.get_sfactor(opts)), float(g["uvs_sfactor"]), rtol=1e-5)
    _canvas_close(full, g["canvas_level2_clear_uvsmap"])
    with pytest.raises(RuntimeError):
        painting.PaintingHelper(ops).render_tiles(g["geom_padded"], g["crops"][:1], opts)


def test_on_white_matches_reference_formula(eng):
    """--on_white (paint_image_main.py:179-183): float32 compositing over white, 3 channels out."""
    helper, (out, full, crops, padded) = _paint(eng, 2)
    ops = OracleTileOps(eng["cfg"], eng["sd"], eng["esd"])
    h2 = painting.PaintingHelper(ops, batch=4)
    h2.set_feature_blending(2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(eng["z"]), 594)
    m = int(eng["g"]["crop_margin"])
    white = h2.paint_image(eng["g"]["geom"], opts, crop_margin=m, on_white=True)
    alpha = full[..., 3:].astype(np.float32) / 255
    ref = (full[..., :3].astype(np.float32) * alpha + 255 * (1 - alpha)).clip(0, 255).astype(np.uint8)
    h0, w0 = eng["g"]["geom"].shape
    assert np.array_equal(white, ref[m:m + h0, m:m + w0]) and white.shape == (h0, w0, 3)
