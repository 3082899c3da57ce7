"""Generate golden vectors by running the REFERENCE implementation (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference (read-only, never shipped), builds the reference ``Generator`` with this
repo's seeded synthetic weights loaded through ``load_state_dict(strict=True)``, runs it on CPU in
fp32 (``force_fp32=True``, ``noise_mode='const'``; the custom ops take their ``_ref`` path on CPU,
bias_act.py:87-89, upfirdn2d.py:162-164) and stores inputs + outputs as small ``.npz`` fixtures next
to this file.  The fixtures are data only.  The tests that consume them never read /root/reference.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import thirdparty.stylegan2_ada_pytorch  # noqa: E402  (puts torch_utils/dnnlib/training on sys.path)
import thirdparty.stylegan2_ada_pytorch.dnnlib as dnnlib  # noqa: E402
from thirdparty.stylegan2_ada_pytorch.training.networks_modified import Generator  # noqa: E402
from thirdparty.stylegan2_ada_pytorch.training import networks as ref_networks  # noqa: E402
from torch_utils.ops import bias_act as ref_bias_act  # noqa: E402
from torch_utils.ops import upfirdn2d as ref_upfirdn2d  # noqa: E402

# forger.train.stitching imports torchvision (unused there); stub it so BlendedFeatures can be imported
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
from forger.train.stitching import BlendedFeatures  # noqa: E402

from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic  # noqa: E402


def build_reference(cfg, sd):
    G = Generator(z_dim=cfg.z_dim, c_dim=0, w_dim=cfg.w_dim, img_resolution=cfg.img_resolution, img_channels=3,
                  mapping_kwargs=dnnlib.EasyDict(num_layers=cfg.mapping_layers),
                  synthesis_kwargs=dnnlib.EasyDict(
                      channel_base=cfg.channel_base, channel_max=cfg.channel_max, num_fp16_res=0,
                      conv_clamp=cfg.conv_clamp, architecture="orig", color_format="triad", color_w_channels=0,
                      enable_geom_linear=False, geom_feature_channels=list(cfg.geom_feature_channels),
                      geom_feature_resolutions=list(cfg.geom_feature_resolutions))).eval().requires_grad_(False)
    ref_sd = G.state_dict()
    # our noise_grid / resample_filter formulas must reproduce the reference's own buffers bit for bit
    for k, v in ref_sd.items():
        if k.endswith("noise_grid") or k.endswith("resample_filter"):
            assert np.array_equal(v.numpy(), sd[k]), k
    G.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert G.num_ws == cfg.num_ws
    return G


class Capture:
    """Forward hooks on every SynthesisLayer + a softmax spy for the ToRGB logits."""

    def __init__(self, G, cfg):
        self.out = {}
        self.handles = []
        for l in cfg.layers:
            mod = G
            for part in l.name.split("."):
                mod = getattr(mod, part)
            self.handles.append(mod.register_forward_hook(self._hook(l.name)))
        self._softmax = torch.softmax

    def _hook(self, name):
        def fn(mod, inp, out):
            self.out[f"{name}.out"] = out.detach().clone()
        return fn

    def __enter__(self):
        def spy(x, dim):
            self.out["torgb.logits"] = x.detach().clone()
            return self._softmax(x, dim=dim)
        torch.softmax = spy
        return self

    def __exit__(self, *a):
        torch.softmax = self._softmax
        for h in self.handles:
            h.remove()


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def sub(a, step):
    """Strided subsample of the two spatial axes (keeps fixtures small)."""
    return np.ascontiguousarray(a[..., ::step, ::step])


def stats(a):
    a = np.asarray(a, np.float64)
    return np.array([a.sum(), (a * a).sum(), np.abs(a).max()], np.float64)


# ----------------------------------------------------------------------------------------------

def make_ops():
    torch.manual_seed(0)                  # (the FC known-answer tests draw their weights from torch's generator)
    rs = np.random.RandomState(1234)
    out = {}
    # bias_act KATs (bias_act.py:93-123): lrelu / linear / tanh x clamp on/off
    x = (rs.randn(3, 5, 6, 7) * 3).astype(np.float32)
    b = rs.randn(5).astype(np.float32)
    out["ba_x"], out["ba_b"] = x, b
    for act in ("lrelu", "linear", "tanh"):
        for clamp in (None, 1.5):
            y = ref_bias_act.bias_act(torch.from_numpy(x), torch.from_numpy(b), act=act, clamp=clamp, impl="ref")
            out[f"ba_{act}_{'c' if clamp else 'n'}"] = np32(y)
    y = ref_bias_act.bias_act(torch.from_numpy(x), torch.from_numpy(b), act="lrelu", gain=np.sqrt(2) * 0.5,
                              clamp=256 * 0.5, impl="ref")
    out["ba_lrelu_gain"] = np32(y)
    x2 = rs.randn(4, 9).astype(np.float32)
    b2 = rs.randn(9).astype(np.float32)
    out["ba2_x"], out["ba2_b"] = x2, b2
    out["ba2_tanh"] = np32(ref_bias_act.bias_act(torch.from_numpy(x2), torch.from_numpy(b2), dim=1, act="tanh", impl="ref"))

    # upfirdn2d KATs (upfirdn2d.py:168-208)
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    out["fir_f"] = np32(f)
    xu = rs.randn(2, 3, 9, 9).astype(np.float32)
    out["fir_x"] = xu
    out["fir_pad1_gain4"] = np32(ref_upfirdn2d.upfirdn2d(torch.from_numpy(xu), f, padding=[1, 1, 1, 1], gain=4, impl="ref"))
    out["fir_up2"] = np32(ref_upfirdn2d.upfirdn2d(torch.from_numpy(xu), f, up=2, padding=[2, 1, 2, 1], gain=4, impl="ref"))
    out["fir_down2"] = np32(ref_upfirdn2d.upfirdn2d(torch.from_numpy(xu), f, down=2, padding=[1, 1, 1, 1], impl="ref"))
    frag = torch.from_numpy(rs.rand(3, 2).astype(np.float32))
    out["fir_f_ragged"] = np32(frag)
    out["fir_ragged_flip"] = np32(ref_upfirdn2d.upfirdn2d(torch.from_numpy(xu), frag, padding=[1, 0, 2, -1], flip_filter=True, gain=1.5, impl="ref"))

    # modulated_conv2d KATs (networks.py:30-88)
    n, ic, oc, h = 2, 6, 5, 7
    x = rs.randn(n, ic, h, h).astype(np.float32)
    w = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    out["mc_x"], out["mc_w"], out["mc_s"] = x, w, s
    for up in (1, 2):
        noise = rs.randn(n, 1, h * up, h * up).astype(np.float32)
        out[f"mc_noise_up{up}"] = noise
        for demod in (True, False):
            for use_noise in (True, False):
                for fused in (True, False):
                    y = ref_networks.modulated_conv2d(
                        torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(s),
                        noise=torch.from_numpy(noise) if use_noise else None, up=up, padding=1,
                        resample_filter=f, demodulate=demod, flip_weight=(up == 1), fused_modconv=fused)
                    out[f"mc_up{up}_d{int(demod)}_n{int(use_noise)}_f{int(fused)}"] = np32(y)
    w1 = rs.randn(3, ic, 1, 1).astype(np.float32)
    out["mc_w1x1"] = w1
    out["mc_1x1_nodemod"] = np32(ref_networks.modulated_conv2d(
        torch.from_numpy(x), torch.from_numpy(w1), torch.from_numpy(s), demodulate=False))

    # FullyConnectedLayer (networks.py:109-122) and normalize_2nd_moment (:24-26)
    fc = ref_networks.FullyConnectedLayer(8, 5, activation="lrelu", lr_multiplier=0.01)
    fcl = ref_networks.FullyConnectedLayer(8, 5, activation="linear", bias_init=1)
    xin = rs.randn(3, 8).astype(np.float32)
    with torch.no_grad():
        fc.bias.copy_(torch.from_numpy(rs.randn(5).astype(np.float32) * 10))
    out["fc_x"] = xin
    out["fc_lrelu_w"], out["fc_lrelu_b"] = np32(fc.weight), np32(fc.bias)
    out["fc_lrelu_y"] = np32(fc(torch.from_numpy(xin)))
    out["fc_lin_w"], out["fc_lin_b"] = np32(fcl.weight), np32(fcl.bias)
    out["fc_lin_y"] = np32(fcl(torch.from_numpy(xin)))
    out["n2m_y"] = np32(ref_networks.normalize_2nd_moment(torch.from_numpy(xin)))

    # position-shifted constant noise (networks.py:373-381), r = 8 with image resolution 32
    r, R = 8, 32
    noise_const = rs.randn(r, r).astype(np.float32)
    grid = ref_networks.create_sampling_grid(r)
    pos = torch.tensor([[0, 0], [37, 211], [31, 1], [5, 30], [16, 16]], dtype=torch.int64)
    normp = (pos % R) / (R - 1)
    ns = torch.nn.functional.grid_sample(
        torch.from_numpy(noise_const)[None, None].expand(pos.shape[0], -1, -1, -1),
        ((grid + normp.unsqueeze(1).unsqueeze(1)) % 1) * 2 - 1, padding_mode="reflection", align_corners=True)
    out["ns_noise"], out["ns_grid"], out["ns_pos"], out["ns_R"] = noise_const, np32(grid), pos.numpy(), np.int64(R)
    out["ns_out"] = np32(ns)
    np.savez_compressed(os.path.join(HERE, "ops_kat.npz"), **out)
    print("ops_kat.npz:", len(out), "arrays")


def run_case(G, cfg, cap, z=None, ws=None, geom=None, positions=None, **kw):
    cap.out.clear()
    gf = [torch.from_numpy(g) for g in geom]
    pos = None if positions is None else torch.from_numpy(positions)
    with torch.no_grad():
        if ws is None:
            res = G(z=torch.from_numpy(z), c=None, geom_feature=gf, positions=pos, noise_mode="const",
                    force_fp32=True, **kw)
        else:
            res = G.forward_pre_mapped(ws=torch.from_numpy(ws), geom_feature=gf, positions=pos, noise_mode="const",
                                       force_fp32=True, **kw)
    return res, dict(cap.out)


def make_tiny():
    cfg = cfgmod.tiny_config(32)
    sd = wmod.random_state_dict(cfg, seed=11)
    G = build_reference(cfg, sd)
    n = 3
    z = synthetic.batch_z(cfg, n, first_seed=594)
    geom = synthetic.geom_features(cfg, n, seed=3)
    pos = np.array([[0, 0], [37, 211], [4095, 17]], np.int64)
    out = {"z": z, "positions": pos, "weights_seed": np.int64(11), "geom_seed": np.int64(3)}
    half = cfg.img_resolution // 2
    with Capture(G, cfg) as cap:
        # case A: z entry, positions, debug data + features at R/2 (the stylize path, BR:731-761)
        (img, dbg), taps = run_case(G, cfg, cap, z=z, geom=geom, positions=pos, return_debug_data=True,
                                    return_features=[half])
        out["A_img"], out["A_uvs"], out["A_colors"], out["A_ws"] = np32(img), np32(dbg["uvs"]), np32(dbg["colors"]), np32(dbg["ws"])
        out[f"A_features{half}"] = np32(dbg[f"features{half}"])
        out[f"A_features{half}_preblend"] = np32(dbg[f"features{half}_preblend"])
        for k, v in taps.items():
            out[f"A_{k}"] = np32(v)
        # case B: no positions, plain return
        img, _ = run_case(G, cfg, cap, z=z, geom=geom, positions=None)
        out["B_img"] = np32(img)
        # case C: W+ entry with per-layer ws and noise buffer overrides (BR:746-754, NM:163-165)
        rs = np.random.RandomState(5)
        ws = (np32(dbg["ws"]) + 0.3 * rs.randn(n, cfg.num_ws, cfg.w_dim)).astype(np.float32)
        nbuf = {"b8.conv0.noise_const": rs.randn(8, 8).astype(np.float32),
                "b32.conv1.noise_const": rs.randn(32, 32).astype(np.float32)}
        out["C_ws"] = ws
        for k, v in nbuf.items():
            out[f"C_nb_{k}"] = v
        (img, dbg), _ = run_case(G, cfg, cap, ws=ws, geom=geom, positions=pos[::-1].copy(), return_debug_data=True,
                                 noise_buffers={k: torch.from_numpy(v) for k, v in nbuf.items()})
        out["C_img"], out["C_uvs"], out["C_colors"] = np32(img), np32(dbg["uvs"]), np32(dbg["colors"])
        # case D: feature blending at R/2 (NM:179-185) for a single patch, broadcast canvas features
        feat = rs.randn(1, cfg.channels(half), half, half).astype(np.float32)
        alpha = rs.rand(1, 1, half, half).astype(np.float32)
        alpha[..., : half // 4, :] = 0
        alpha[..., -half // 4:, :] = 1
        out["D_feat"], out["D_alpha"] = feat, alpha
        bf = {half: BlendedFeatures(torch.from_numpy(feat), torch.from_numpy(alpha))}
        (img, dbg), _ = run_case(G, cfg, cap, z=z[:1], geom=[g[:1] for g in geom], positions=pos[:1],
                                 return_debug_data=True, return_features=[half], blended_features=bf)
        out["D_img"], out["D_uvs"] = np32(img), np32(dbg["uvs"])
        out[f"D_features{half}"], out[f"D_features{half}_preblend"] = np32(dbg[f"features{half}"]), np32(dbg[f"features{half}_preblend"])
        # case E: blending at the LAST resolution (torgb is redone, NM:182-185) + truncation psi
        R = cfg.img_resolution
        featR = rs.randn(1, cfg.channels(R), R, R).astype(np.float32)
        alphaR = rs.rand(1, 1, R, R).astype(np.float32)
        out["E_feat"], out["E_alpha"] = featR, alphaR
        bf = {R: BlendedFeatures(torch.from_numpy(featR), torch.from_numpy(alphaR))}
        (img, dbg), _ = run_case(G, cfg, cap, z=z[:1], geom=[g[:1] for g in geom], positions=None,
                                 return_features=[R], blended_features=bf, truncation_psi=0.7)
        out["E_img"], out["E_uvs"] = np32(img), np32(dbg["uvs"])
        out[f"E_features{R}"] = np32(dbg[f"features{R}"])
    np.savez_compressed(os.path.join(HERE, "gen_tiny.npz"), **out)
    print("gen_tiny.npz:", len(out), "arrays")


def make_full(res, step):
    cfg = cfgmod.style1_config(res)
    sd = wmod.random_state_dict(cfg, seed=0)
    G = build_reference(cfg, sd)
    n = 2
    z = synthetic.batch_z(cfg, n, first_seed=594)
    geom = synthetic.geom_features(cfg, n, seed=0)
    pos = np.array([[0, 0], [37, 211]], np.int64)
    out = {"z": z, "positions": pos, "weights_seed": np.int64(0), "geom_seed": np.int64(0), "step": np.int64(step)}
    half = res // 2
    with Capture(G, cfg) as cap:
        (img, dbg), taps = run_case(G, cfg, cap, z=z, geom=geom, positions=pos, return_debug_data=True,
                                    return_features=[half])
    out["ws"], out["colors"] = np32(dbg["ws"]), np32(dbg["colors"])
    full = {"img": np32(img), "uvs": np32(dbg["uvs"]), "logits": np32(taps["torgb.logits"]),
            f"features{half}": np32(dbg[f"features{half}"])}
    for k, v in taps.items():
        if k.endswith(".out"):
            full[k] = np32(v)
    for k, v in full.items():
        out[f"{k}.stats"] = stats(v)                       # sum, sum of squares, max-abs over the FULL tensor
        s = 1 if v.shape[-1] <= 16 else step
        out[f"{k}.sub"] = sub(v, s) if v.shape[1] <= 4 else sub(v[:, ::8], s)   # every 8th channel for features
    out["uvs.row"] = full["uvs"][:, :, res // 3, :]      # one full row of pixels
    np.savez_compressed(os.path.join(HERE, f"gen_r{res}.npz"), **out)
    print(f"gen_r{res}.npz:", len(out), "arrays")


def make_hdr():
    """High-dynamic-range fixture (R=128, N=2): weights.hdr_state_dict drives several layers into conv_clamp = 256 and
    the triad logits over +-20; full uvs / img of the reference's fp32 evaluation + per-layer range statistics."""
    res = 128
    cfg = cfgmod.style1_config(res)
    sd = wmod.hdr_state_dict(cfg, seed=0)
    G = build_reference(cfg, sd)
    n = 2
    z = synthetic.batch_z(cfg, n, first_seed=594)
    geom = synthetic.geom_features(cfg, n, seed=0)
    pos = np.array([[0, 0], [37, 211]], np.int64)
    out = {"z": z, "positions": pos, "weights_seed": np.int64(0), "geom_seed": np.int64(0)}
    with Capture(G, cfg) as cap:
        (img, dbg), taps = run_case(G, cfg, cap, z=z, geom=geom, positions=pos, return_debug_data=True,
                                    return_features=[res // 2])
    out["uvs"], out["img"], out["colors"] = np32(dbg["uvs"]), np32(img), np32(dbg["colors"])
    out["logits.sub"] = sub(np32(taps["torgb.logits"]), 4)
    out[f"features{res // 2}.sub"] = sub(np32(dbg[f"features{res // 2}"])[:, ::8], 4)
    names, rng = [], []
    for l in cfg.layers:
        v = np32(taps[f"{l.name}.out"])
        names.append(l.name)
        rng.append([np.sqrt((v.astype(np.float64) ** 2).mean()), np.abs(v).max(), (np.abs(v) >= 255.99).mean()])
    out["layer_range"] = np.array(rng, np.float64)               # rms, max-abs, fraction at the clamp, per layer
    lg = np32(taps["torgb.logits"])
    out["logits.range"] = np.array([lg.min(), lg.max()], np.float64)
    assert (out["layer_range"][:, 1] >= 255.99).sum() >= 5 and lg.min() < -15 and lg.max() > 20
    np.savez_compressed(os.path.join(HERE, "gen_hdr_r128.npz"), **out)
    print("gen_hdr_r128.npz: layers at the clamp:", [n_ for n_, r in zip(names, rng) if r[1] >= 255.99], "logits", out["logits.range"])


def make_trained():
    """Trained-like statistics (weights.trained_like_state_dict: log-normal channel scales, dominant styles, strong noise),
    R=128, N=6 (the batch from which the 64 x 64 layers take the large-tile kernels, i.e. run in the mode's own arithmetic)
    with different latents and positions: the reference's fp32 evaluation -- full uvs / img, the features at R/2 (what the
    FeatureCanvas carries from stroke to stroke) on every 8th channel, logits subsampled, per-layer ranges."""
    res = 128
    cfg = cfgmod.style1_config(res)
    sd = wmod.trained_like_state_dict(cfg, seed=0)
    G = build_reference(cfg, sd)
    n = 6
    z = synthetic.batch_z(cfg, n, first_seed=1234)
    geom = synthetic.geom_features(cfg, n, seed=3)
    pos = np.array([[0, 0], [37, 211], [4095, 17], [5, 5], [100, 3], [77, 900]], np.int64)
    out = {"z": z, "positions": pos, "weights_seed": np.int64(0), "geom_seed": np.int64(3)}
    half = res // 2
    with Capture(G, cfg) as cap:
        (img, dbg), taps = run_case(G, cfg, cap, z=z, geom=geom, positions=pos, return_debug_data=True, return_features=[half])
    out["uvs"], out["img.sub"], out["colors"] = np32(dbg["uvs"]), sub(np32(img), 2), np32(dbg["colors"])
    out["logits.sub"] = sub(np32(taps["torgb.logits"]), 2)
    f = np32(dbg[f"features{half}"])
    out[f"features{half}.c16"] = f[:, ::16]
    out[f"features{half}.maxabs"] = np.float64(np.abs(f).max())
    rng = []
    for l in cfg.layers:
        v = np32(taps[f"{l.name}.out"])
        rng.append([np.sqrt((v.astype(np.float64) ** 2).mean()), np.abs(v).max(), (np.abs(v) >= 255.99).mean()])
    out["layer_range"] = np.array(rng, np.float64)
    lg = np32(taps["torgb.logits"])
    out["logits.range"] = np.array([lg.min(), lg.max()], np.float64)
    np.savez_compressed(os.path.join(HERE, "gen_trained_r128.npz"), **out)
    print("gen_trained_r128.npz: layer rms/max/clamped", np.round(out["layer_range"], 3).tolist(), "logits", out["logits.range"],
          "features max", out[f"features{half}.maxabs"])


def make_b32():
    """The BASELINE workload itself (batch 32, R=256, the inputs bench.py's rank 0 uses) through the reference on CPU:
    per-sample checksums of uvs / img and one full pixel row per sample."""
    res, n = 256, 32
    cfg = cfgmod.style1_config(res)
    sd = wmod.random_state_dict(cfg, seed=0)
    G = build_reference(cfg, sd)
    z = synthetic.batch_z(cfg, n, first_seed=0)
    geom = synthetic.geom_features(cfg, n, seed=0)
    pos = synthetic.positions(cfg, n, seed=0)
    uvs, img, colors = [], [], []
    with torch.no_grad():
        for i in range(0, n, 8):
            im, dbg = G(z=torch.from_numpy(z[i:i + 8]), c=None, geom_feature=[torch.from_numpy(g[i:i + 8]) for g in geom],
                        positions=torch.from_numpy(pos[i:i + 8]), noise_mode="const", force_fp32=True, return_debug_data=True)
            uvs.append(np32(dbg["uvs"])); img.append(np32(im)); colors.append(np32(dbg["colors"]))
    uvs, img, colors = np.concatenate(uvs), np.concatenate(img), np.concatenate(colors)
    u64, i64 = uvs.astype(np.float64), img.astype(np.float64)
    out = {"weights_seed": np.int64(0), "first_seed": np.int64(0), "geom_seed": np.int64(0), "pos_seed": np.int64(0),
           "colors": colors,
           "uvs.sum": u64.sum(axis=(2, 3)), "uvs.sumsq": (u64 * u64).sum(axis=(2, 3)),          # [32, 3]
           "img.sum": i64.sum(axis=(2, 3)), "img.sumsq": (i64 * i64).sum(axis=(2, 3)),
           "uvs.row": uvs[:, :, 85, :], "img.row": img[:, :, 170, :], "uvs.sub": uvs[:, :, ::32, ::32]}
    np.savez_compressed(os.path.join(HERE, "gen_b32_r256.npz"), **out)
    print("gen_b32_r256.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--hdr" in sys.argv or "--b32" in sys.argv or "--trained" in sys.argv:
        if "--trained" in sys.argv:
            make_trained()
        if "--hdr" in sys.argv:
            make_hdr()
        if "--b32" in sys.argv:
            make_b32()
        sys.exit(0)
    make_ops()
    make_tiny()
    make_full(128, 4)
    make_full(256, 8)
    make_hdr()
    make_trained()
    make_b32()
