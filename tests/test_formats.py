"""Formats row: websocket binary protocol, brush libraries, engine snapshot container.  The cross-checks
against the reference's own functions run only where /root/reference exists (the build container)."""
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import REPO
from brushstroke_engine_amd import config as cfgmod, encoder as encmod, formats, painting, weights as wmod

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _request(w, h, x, y, m, rs):
    stroke = rs.randint(0, 256, (h, w, 4)).astype(np.uint8)
    return np.array([w, h, x, y, m], np.int32).tobytes() + stroke.tobytes() + stroke[::-1].tobytes(), stroke


def test_protocol_roundtrip():
    rs = np.random.RandomState(0)
    msg, stroke = _request(8, 6, 120, -4, 10, rs)
    head = bytes([1, 2, 7]) + bytes([0, 255, 0, 0]) + bytes([1, 0, 128, 255])
    meta, off = formats.decode_render_request_metadata(head + msg)
    assert meta["debug"] is True and meta["extra_data"] == 7 and off == 11
    assert [list(c) for c in meta["colors"]] == [[0, 255, 0, 0], [1, 0, 128, 255]]
    m, img, canvas = formats.binary_to_image_patches(head + msg, offset=off)
    assert m == {"width": 8, "height": 6, "x": 120, "y": -4, "crop_margin": 10} and canvas is None
    assert np.array_equal(img, stroke)
    out = formats.image_patch_to_binary(stroke, 5, 9)
    assert list(np.frombuffer(out, np.int32, 4)) == [8, 6, 5, 9] and out[16:] == stroke.tobytes()
    assert formats.int32_to_binary(-3) == np.array([-3], np.int32).tobytes()
    with pytest.raises(RuntimeError):
        formats.image_patch_to_binary(stroke.astype(np.float32), 0, 0)
    with pytest.raises(ValueError):
        formats.binary_to_image_patches(msg[:40])


@needs_ref
def test_protocol_matches_reference():
    sys.path.insert(0, REF)
    # import stubs for packages the container lacks (nothing of them is exercised by the three functions compared)
    for name in ("tornado", "tornado.websocket", "tornado.web", "tornado.gen", "skimage", "skimage.io", "skimage.filters",
                 "torchvision"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["tornado.websocket"].WebSocketHandler = object
    sys.modules["tornado"].gen = sys.modules["tornado.gen"]
    sys.modules["tornado.gen"].coroutine = lambda f: f
    sys.modules["skimage.io"].imread = sys.modules["skimage.io"].imsave = None
    sys.modules["skimage.filters"].threshold_otsu = sys.modules["skimage.filters"].threshold_local = None
    import matplotlib
    matplotlib.use("Agg")
    import thirdparty.stylegan2_ada_pytorch  # noqa: F401
    try:
        import forger.ui.util as ref
    except Exception as e:                                    # other optional dependencies of the UI module
        pytest.skip(f"reference ui module not importable here: {e}")
    rs = np.random.RandomState(1)
    msg, stroke = _request(16, 16, 3, 4, 2, rs)
    a, b = ref.binary_to_image_patches(msg), formats.binary_to_image_patches(msg)
    assert {k: int(v) for k, v in a[0].items()} == b[0] and np.array_equal(a[1], b[1])
    assert ref.image_patch_to_binary(stroke, 7, 8) == formats.image_patch_to_binary(stroke, 7, 8)
    head = bytes([0, 1, 3, 2, 9, 8, 7])
    ra, rb = ref.decode_render_request_metadata(head), formats.decode_render_request_metadata(head)
    assert ra[1] == rb[1] and list(ra[0]["colors"][0]) == list(rb[0]["colors"][0])


def test_seed_and_w_libraries(tmp_path):
    f = tmp_path / "seeds.txt"
    f.write_text("# my brushes\n594 0.1 0.2\n\n12\nbroken line\n7 1 2 3\n")
    zs, zdim = formats.read_zs(str(f))
    assert zs == [594, 12, 7] and zdim == 3
    lib = formats.BrushLibrary.from_arg(str(f), z_dim=64)
    assert isinstance(lib, formats.SeedBrushLibrary) and lib.get_style_ids() == ["12", "594", "7"]
    opts = painting.GanBrushOptions()
    lib.set_style("594", opts)
    assert opts.style_id == "594" and opts.style_ws is None
    assert np.array_equal(opts.style_z.numpy(), np.random.RandomState(594).randn(1, 64))
    rl = formats.BrushLibrary.from_arg("rand5", 64)
    assert rl.get_style_ids() == ["rand0", "rand1", "rand2", "rand3", "rand4"]
    rl.set_style("rand3", opts)
    g = torch.Generator(); g.manual_seed(1)
    assert torch.equal(opts.style_z, torch.rand((1, 64), generator=g)) and opts.style_id is None
    assert formats.BrushLibrary.from_arg("3,1,2").get_style_ids() == ["1", "2", "3"]
    assert len(formats.BrushLibrary.from_arg("4").zs) == 4
    lib.set_interpolated_style("594", "7", 0.25, opts)
    assert opts.style_id == "594_0.25__7"
    # W library: bare ws, {'w', 'noise'}, and the legacy flat dict
    ws = lambda s: torch.from_numpy(np.random.RandomState(s).randn(1, 14, 64).astype(np.float32))
    noise = lambda s: {"b8.conv0.noise_const": np.random.RandomState(s).randn(8, 8).astype(np.float32)}
    styles = {"a": ws(1), "b": {"w": ws(2), "noise": noise(2)}, "c": dict(w=ws(3), **noise(3))}
    wf = tmp_path / "proj.pkl"
    wf.write_bytes(pickle.dumps(styles))
    wl = formats.BrushLibrary.from_file(str(wf))
    assert isinstance(wl, formats.WBrushLibrary) and wl.get_style_ids() == ["a", "b", "c"]
    wl.set_style("a", opts)
    assert opts.style_z is None and torch.equal(opts.style_ws, styles["a"]) and opts.custom_args == {"noise_buffers": None}
    wl.set_style("c", opts)
    assert torch.is_tensor(opts.custom_args["noise_buffers"]["b8.conv0.noise_const"])
    wl.set_interpolated_style("b", "c", 0.5, opts)
    want = 0.5 * noise(2)["b8.conv0.noise_const"] + 0.5 * noise(3)["b8.conv0.noise_const"]
    np.testing.assert_allclose(opts.custom_args["noise_buffers"]["b8.conv0.noise_const"].numpy(), want, rtol=1e-6)
    np.testing.assert_allclose(opts.style_ws.numpy(), (0.5 * ws(2) + 0.5 * ws(3)).numpy(), rtol=1e-6)


def test_engine_snapshot_roundtrip(tmp_path):
    cfg = cfgmod.tiny_config(32)
    sd = wmod.random_state_dict(cfg, seed=3)
    esd = encmod.random_encoder_state_dict(4)
    p = str(tmp_path / "engine.npz")
    formats.save_engine_snapshot(p, cfg, sd, esd, preproc_type="-11inverse", extra={"color_format": "triad"})
    cfg2, sd2, esd2, pre, extra = formats.load_engine_snapshot(p)
    assert cfg2 == cfg and pre == "-11inverse" and extra == {"color_format": "triad"}
    assert sd.keys() == sd2.keys() and all(np.array_equal(sd[k], sd2[k]) for k in sd)
    assert esd.keys() == esd2.keys() and all(np.array_equal(esd[k], esd2[k]) for k in esd)
    wmod.validate_state_dict(cfg2, sd2)


@needs_ref
def test_convert_reference_snapshot(tmp_path):
    """tools/convert_snapshot.py on a snapshot pickled from the reference's own classes."""
    mk = tmp_path / "mk.py"
    mk.write_text(f'''
import sys, types, argparse, pickle, numpy as np, torch
sys.dont_write_bytecode = True
sys.path.insert(0, {REPO!r}); sys.path.insert(0, {REF!r})
import thirdparty.stylegan2_ada_pytorch
import thirdparty.stylegan2_ada_pytorch.dnnlib as dnnlib
from thirdparty.stylegan2_ada_pytorch.training.networks_modified import Generator
from thirdparty.stylegan2_ada_pytorch.training.networks import Discriminator
import forger.experimental.autoenc.simple_autoencoder as sa
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod
cfg = cfgmod.style1_config(128); sd = wmod.random_state_dict(cfg, seed=0)
G = Generator(z_dim=64, c_dim=0, w_dim=64, img_resolution=128, img_channels=3, mapping_kwargs=dnnlib.EasyDict(num_layers=4),
              synthesis_kwargs=dnnlib.EasyDict(channel_base=16384, channel_max=128, num_fp16_res=0, conv_clamp=256, architecture="orig",
                                               color_format="triad", color_w_channels=0, enable_geom_linear=False,
                                               geom_feature_channels=[16, 256], geom_feature_resolutions=[16, 32])).eval().requires_grad_(False)
G.load_state_dict({{k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, strict=True)
ap = argparse.ArgumentParser(); sa.add_model_flags(ap); ea = ap.parse_args([]); ea.preproc_type = "inverse"
esd = {{k: torch.from_numpy(v) for k, v in encmod.random_encoder_state_dict(5).items()}}
D = Discriminator(c_dim=0, img_resolution=128, img_channels=3, architecture="orig", channel_base=16384, channel_max=128)
snap = dict(G=G, D=D, G_ema=G, training_set_kwargs=None, augment_pipe=None,
            args=argparse.Namespace(color_format="triad", geom_inject_resolutions=[0, 1]), encoder={{"args": ea, "model_state": esd}})
pickle.dump(snap, open({str(tmp_path / "snap.pkl")!r}, "wb"))
''')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, str(mk)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = str(tmp_path / "engine.npz")
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "convert_snapshot.py"), "--reference", REF,
                           "--pkl", str(tmp_path / "snap.pkl"), "--out", out], env=env, stdout=subprocess.DEVNULL)
    cfg, sd, esd, pre, extra = formats.load_engine_snapshot(out)
    assert cfg == cfgmod.style1_config(128) and pre == "inverse" and extra["geom_inject_resolutions"] == [0, 1]
    want = wmod.random_state_dict(cfg, seed=0)
    assert want.keys() == sd.keys() and all(np.array_equal(np.asarray(want[k]), sd[k]) for k in want)
    assert all(np.array_equal(encmod.random_encoder_state_dict(5)[k], esd[k]) for k in esd)
