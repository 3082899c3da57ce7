"""GPU parity of the hand-written geometry encoder (csrc/nb_encoder.hip) against the reference's encoder outputs
(tests/golden/engine_r128.npz) and the oracle restatement on seeded inputs; tolerance 2e-4 on O(1) feature values
(split-f16 products + BatchNorm folding; the generator's pixel budget is 1e-3)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from brushstroke_engine_amd import encoder as encmod, _lib
from oracle import painting_oracle as po

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _p(t):
    return t.data_ptr()


def _h2_to_nchw(t):
    n, c8, _, h, w, _ = t.shape
    v = t.float()
    return (v[:, :, 0] + v[:, :, 1]).permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)


def _nchw_to_h2(x):
    n, c, h, w = x.shape
    hi = x.half()
    lo = (x - hi.float()).half()
    return torch.stack([hi, lo], 1).reshape(n, 2, c // 8, 8, h, w).permute(0, 2, 1, 4, 5, 3).contiguous()


def test_encoder_matches_reference_golden():
    g = load_golden("engine_r128.npz")
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(int(g["encoder_seed"])))
    f = enc.encode(torch.from_numpy(g["enc_in"]).cuda())
    np.testing.assert_allclose(f[0].cpu().numpy(), g["enc_f0"], atol=TOL)
    np.testing.assert_allclose(f[1].cpu().numpy()[:, ::8], g["enc_f1"], atol=TOL)


@pytest.mark.parametrize("res,n,pre", [(128, 3, None), (256, 2, None), (128, 1, "-11inverse"), (256, 1, "inverse"), (384, 2, None)])
def test_encoder_matches_oracle(res, n, pre):
    rs = np.random.RandomState(res + n)
    esd = encmod.random_encoder_state_dict(11)
    geom = (rs.rand(n, 1, res, res) > 0.1).astype(np.float32)
    geom[:, :, ::7, :] = rs.rand(n, 1, len(range(0, res, 7)), res).astype(np.float32)      # gray levels too
    enc = encmod.HipGeometryEncoder(esd, preproc_type=pre)
    f = enc.encode(torch.from_numpy(geom).cuda())
    ref = po.encoder_encode(esd, torch.from_numpy(geom), pre)
    assert f[0].shape == ref[0].shape and f[1].shape == ref[1].shape
    for a, b in zip(f, ref):
        assert float((a.cpu() - b).abs().max()) <= TOL, float((a.cpu() - b).abs().max())


@pytest.mark.parametrize("stride,ci,co,h,w,h2out", [(1, 16, 256, 32, 32, False), (2, 64, 128, 64, 64, True),
                                                     (2, 256, 256, 32, 32, True), (1, 32, 16, 16, 16, False),
                                                     (1, 256, 32, 16, 16, True), (2, 128, 256, 64, 128, False),
                                                     # 48-wide outputs (the inner layers of a 384 x 384 patch): 16-wide tiles, three per row
                                                     (2, 32, 48, 96, 96, True), (1, 32, 16, 48, 48, False)])
@pytest.mark.parametrize("small", [0, 1])
def test_enc_conv_layer(stride, ci, co, h, w, h2out, small):
    """One layer against torch fp64 (reflect pad, cross-correlation), both output formats, both tile shapes, and both
    kernels: the 128 c_out x 256 pixel tiles (small=0) and the 32 x 32 split-K tiles of under-filled launches (small=1)."""
    _lib.lib().nb_debug_set_enc_small(small)
    try:
        _enc_conv_layer_case(stride, ci, co, h, w, h2out)
    finally:
        _lib.lib().nb_debug_set_enc_small(-1)


def _enc_conv_layer_case(stride, ci, co, h, w, h2out):
    rs = np.random.RandomState(ci + co)
    n = 2
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32))
    wt = (rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)
    b = rs.randn(co).astype(np.float32)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(
        torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode="reflect"), torch.from_numpy(wt).double(),
        torch.from_numpy(b).double(), stride=stride), 0.01)
    ho, wo = h // stride, w // stride
    xd = _nchw_to_h2(x).cuda()
    wd = torch.from_numpy(encmod.pack_enc_weight_h3(wt)).cuda()
    bd = torch.from_numpy(b).cuda()
    if h2out:
        y = torch.empty([n, co // 8, 2, ho, wo, 8], dtype=torch.float16, device="cuda")
        _lib.check(_lib.lib().nb_enc_conv3x3_h3(_p(xd), ci, _p(wd), _p(bd), None, _p(y), n, h, w, co, stride, 0.01,
                                                torch.cuda.current_stream().cuda_stream), "enc_conv")
        out = _h2_to_nchw(y).cpu()
    else:
        y = torch.empty([n, co, ho, wo], dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().nb_enc_conv3x3_h3(_p(xd), ci, _p(wd), _p(bd), _p(y), None, n, h, w, co, stride, 0.01,
                                                torch.cuda.current_stream().cuda_stream), "enc_conv")
        out = y.cpu()
    assert float((out.double() - ref).abs().max()) <= 2e-5


def test_enc_stem_and_upsample():
    rs = np.random.RandomState(0)
    n, h, w = 2, 32, 64
    x = torch.from_numpy(rs.rand(n, 1, h, w).astype(np.float32))
    wt = (rs.randn(64, 1, 7, 7) / 7).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(
        torch.nn.functional.pad(x.double(), (3, 3, 3, 3), mode="reflect"), torch.from_numpy(wt).double(),
        torch.from_numpy(b).double()), 0.01)
    w50 = np.zeros([64, 50], np.float32)
    w50[:, :49] = wt.reshape(64, 49)
    y = torch.empty([n, 8, 2, h, w, 8], dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    xd, wd, bd = x.cuda(), torch.from_numpy(w50).cuda(), torch.from_numpy(b).cuda()      # keep the device buffers alive
    _lib.check(_lib.lib().nb_enc_stem7x7_f32_h2(_p(xd), _p(wd), _p(bd), _p(y), n, h, w, 0, 0.01, st), "stem")
    assert float((_h2_to_nchw(y).cpu().double() - ref).abs().max()) <= 1e-5
    xs = torch.from_numpy(rs.randn(n, 16, 16, 32).astype(np.float32))
    up = torch.empty([n, 2, 2, 32, 64, 8], dtype=torch.float16, device="cuda")
    xsd = xs.cuda()
    _lib.check(_lib.lib().nb_enc_upsample2x_h2(_p(xsd), _p(up), n, 16, 16, 32, st), "upsample")
    ref = torch.nn.functional.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=True)
    assert float((_h2_to_nchw(up).cpu() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("res", [64, 32])
@pytest.mark.parametrize("n_rep", [1, 4])
def test_encoder_small_patch_sizes_match_reference(res, n_rep):
    """Patch sizes below 128 (a resolution-generic build): the inner layers' outputs are 8 or 4 pixels wide and
    run on the 32-position split-K tiles (ragged last tile rows masked).  Against the REFERENCE encoder's outputs
    (tests/golden/make_golden_engine.py --encoder-small; simple_autoencoder.py:155-199, 251-261), also at a batch that
    would take the f8 inter-layer format at the large sizes (12 samples: stays on hi/lo f16 here)."""
    g = load_golden("encoder_small.npz")
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(int(g["encoder_seed"])))
    assert enc.supports(res) and not enc.large_tiles_only(res)
    x = torch.from_numpy(np.concatenate([g[f"enc_in_r{res}"]] * n_rep)).cuda()
    f = enc.encode(x)
    for k in range(n_rep):
        np.testing.assert_allclose(f[0][3 * k:3 * k + 3].cpu().numpy(), g[f"enc_f0_r{res}"], atol=TOL)
        np.testing.assert_allclose(f[1][3 * k:3 * k + 3].cpu().numpy(), g[f"enc_f1_r{res}"], atol=TOL)
    lazy = enc.lazy(x)
    assert lazy.feature_shape(1) == (3 * n_rep, 256, res // 4, res // 4)
    assert lazy.can_handoff(1) == (res == 64) and not lazy.can_handoff(0)


@pytest.mark.parametrize("res", [64, 32])
def test_painting_at_small_patch_sizes(res):
    """The engine end to end at patch sizes 64 and 32 (HIP encoder + generator + canvas kernels, feature blending 2, lazy
    geometry provider) against the oracle painter's sequential tile loop."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, painting
    from brushstroke_engine_amd.networks import Generator
    from oracle import neube_oracle as no
    cfg = cfgmod.style1_config(res)
    sd, esd = wmod.random_state_dict(cfg, seed=2), encmod.random_encoder_state_dict(5)
    rs = np.random.RandomState(res)
    geom = np.full((3 * res - 11, 2 * res + 7, 1), 255, np.uint8)
    for _ in range(12):
        y, x = rs.randint(2, geom.shape[0] - 2), rs.randint(2, geom.shape[1] - 2)
        geom[max(y - 9, 0):y + 9, x - 1:x + 2] = 0
        geom[y - 1:y + 2, max(x - 9, 0):x + 9] = 0
    z = np.random.RandomState(594).randn(1, cfg.z_dim)
    m = 4 if res == 32 else 6
    P = po.OraclePainter(no.OracleGenerator(cfg, sd), esd)
    P.feature_blending_margin = 8 if res == 32 else 16          # (16 leaves no interior on the 16 x 16 blending grid of R = 32)
    ref = P.paint_image(geom, z=z, crop_margin=m, feature_blending=2)[0]
    for mode in ("h3", "f8"):
        G = Generator(cfg, sd, conv_mode=mode).to("cuda")
        helper = painting.PaintingHelper(painting.TileOps(G, encmod.HipGeometryEncoder(esd)), batch=5)
        helper.feature_blending_margin = 8 if res == 32 else 16
        helper.set_feature_blending(2)
        opts = painting.GanBrushOptions()
        opts.set_style(torch.from_numpy(z), 594)
        out = helper.paint_image(geom, opts, crop_margin=m)
        d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() <= 5e-3, (res, mode, d.max(), (d > 0).mean())


def test_encoder_rejects_unsupported():
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(1))
    with pytest.raises(RuntimeError):
        enc.encode(torch.zeros(1, 1, 48, 48).cuda())
    with pytest.raises(RuntimeError):
        enc.encode(torch.zeros(1, 1, 64, 32).cuda())
    with pytest.raises(RuntimeError):
        encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(1), preproc_type="bogus")
    assert _lib.lib().nb_enc_conv3x3_h3(None, 16, None, None, None, None, 1, 16, 16, 16, 1, 0.01, None) < 0


@pytest.mark.parametrize("mode", ["f8", "h3"])
def test_lazy_geometry_handoff_equals_fp32_path(mode):
    """HipGeometryEncoder.lazy: the generator asks the encoder to write the 256-channel decoder feature straight into the
    consuming layer's operand tensor (nb_enc_conv3x3_h3_handoff, x the consumer's styles) -- the same pixels as the fp32
    features + packing pass, up to the one rounding the fused scaling saves (compared at the output: uvs)."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.networks import Generator
    cfg = cfgmod.style1_config(256)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode=mode).to("cuda")
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5))
    n = 6
    rs = np.random.RandomState(3)
    geom = torch.from_numpy((rs.rand(n, 1, 256, 256) > 0.1).astype(np.float32)).cuda()
    ws = G.mapping(torch.from_numpy(synthetic.batch_z(cfg, n, 7)).cuda(), None)
    pos = torch.from_numpy(synthetic.positions(cfg, n, seed=1)).cuda()
    _, want = G.forward_pre_mapped(ws, enc.encode(geom), positions=pos, return_debug_data=True, noise_mode="const")
    lazy = enc.lazy(geom)
    _, got = G.forward_pre_mapped(ws, lazy, positions=pos, return_debug_data=True, noise_mode="const")
    assert lazy._plain is None                                    # the fp32 feature tensors were never materialised
    assert float((got["uvs"] - want["uvs"]).abs().max()) <= 2e-5
    # a tapped geometry resolution (features returned at 64) keeps the fp32 route
    _, a = G.forward_pre_mapped(ws, enc.lazy(geom), positions=pos, return_debug_data=True, return_features=[64], noise_mode="const")
    _, b = G.forward_pre_mapped(ws, enc.encode(geom), positions=pos, return_debug_data=True, return_features=[64], noise_mode="const")
    assert torch.equal(a["uvs"], b["uvs"]) and torch.equal(a["features64"], b["features64"])
    # batch 1: no layer takes the hand-off (the producer runs on the small-image kernel): identical to the plain path
    _, c1 = G.forward_pre_mapped(ws[:1], enc.lazy(geom[:1]), positions=pos[:1], return_debug_data=True, noise_mode="const")
    _, c2 = G.forward_pre_mapped(ws[:1], enc.encode(geom[:1]), positions=pos[:1], return_debug_data=True, noise_mode="const")
    assert torch.equal(c1["uvs"], c2["uvs"])


@pytest.mark.parametrize("mode", ["f8", "h3"])
def test_lazy_geometry_batch16_side_stream_ordering(mode):
    """Batch 16 at R=256: b32.conv1 takes the large split-f16 kernel, so BOTH geometry features are handed over in operand
    format -- feature 1 by the encoder (lazy hand-off) and feature 0 by the early pack on the plan's side stream.  That
    pack reads the styles and the encoder's fp32 feature, both just enqueued on the caller's stream: the side stream must
    wait for them although `pre_h2` is already non-empty (a data race of round 2).  A different geometry batch and other
    latents run first, so stale styles / recycled allocator blocks would show up in the result."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.networks import Generator
    cfg = cfgmod.style1_config(256)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode=mode).to("cuda")
    enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5))
    n = 16
    rs = np.random.RandomState(4)
    geoms = [torch.from_numpy((rs.rand(n, 1, 256, 256) > thr).astype(np.float32)).cuda() for thr in (0.5, 0.1)]
    wss = [G.mapping(torch.from_numpy(synthetic.batch_z(cfg, n, sd)).cuda(), None) for sd in (100, 7)]
    pos = torch.from_numpy(synthetic.positions(cfg, n, seed=1)).cuda()
    _, want = G.forward_pre_mapped(wss[1], enc.encode(geoms[1]), positions=pos, return_debug_data=True, noise_mode="const")
    want_uvs = want["uvs"].clone()
    for _ in range(3):
        # the other batch first (fills the workspaces with ITS styles, leaves its buffers in the allocator's free lists)
        G.forward_pre_mapped(wss[0], enc.lazy(geoms[0]), positions=pos, return_debug_data=True, noise_mode="const")
        lazy = enc.lazy(geoms[1])
        _, got = G.forward_pre_mapped(wss[1], lazy, positions=pos, return_debug_data=True, noise_mode="const")
        assert float((got["uvs"] - want_uvs).abs().max()) <= 2e-5


def _f8_decode(t, c):
    """f8-format tensor [n, c/8, 2, h, w, 8] -> (hi + xl) as fp32 NCHW, i.e. the activation to ~15 bits"""
    n, c8, _, h, w, _ = t.shape
    hi = t[:, :, 0].float().permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)
    lo = t[:, :, 1].contiguous().view(torch.uint8).view(torch.float8_e4m3fn).float()      # [n, c8, h, w, 16]
    xl = lo[:, 0::2].permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)
    return (hi + xl / 512)[:, :c]


@pytest.mark.parametrize("stride,ci,co,h,w,h2out", [(1, 16, 256, 32, 32, False), (2, 64, 128, 64, 64, True),
                                                     (2, 256, 256, 32, 32, True), (1, 256, 32, 16, 16, True),
                                                     (2, 128, 256, 64, 128, False), (1, 48, 16, 16, 32, False),
                                                     # stride 2 with per-chunk slabs (round 4): odd chunk counts, ONE chunk, ragged c_out slices,
                                                     # 16-wide tiles, a single tile row
                                                     (2, 48, 136, 32, 64, False), (2, 80, 48, 64, 64, True), (2, 16, 16, 32, 32, True),
                                                     (2, 16, 24, 16, 64, False), (2, 112, 272, 48, 64, True),
                                                     (2, 64, 128, 96, 96, True), (1, 48, 40, 48, 48, False)])
def test_enc_conv_layer_f8(stride, ci, co, h, w, h2out):
    """The same layer with "f8" operands (one f16 + half an fp8 MFMA per tap; the third tap's corrections paired across
    K steps -- odd and even step counts are both here) against torch fp64: the fp8 correction terms leave
    2^-11 * 2^-4 relative per product, i.e. ~20x the hi/lo-f16 error."""
    from brushstroke_engine_amd import ops
    rs = np.random.RandomState(ci + co)
    n = 2
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32))
    wt = (rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)
    b = rs.randn(co).astype(np.float32)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(
        torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode="reflect"), torch.from_numpy(wt).double(),
        torch.from_numpy(b).double(), stride=stride), 0.01)
    ho, wo = h // stride, w // stride
    xd = ops.pack_h2f8(x.cuda(), torch.ones(n, ci, device="cuda"))
    wd = torch.from_numpy(encmod.pack_enc_weight_f8(wt)).cuda()
    bd = torch.from_numpy(b).cuda()
    S = torch.cuda.current_stream().cuda_stream
    if h2out:
        y = torch.zeros([n, co // 8, 2, ho, wo, 8], dtype=torch.float16, device="cuda")
        _lib.check(_lib.lib().nb_enc_conv3x3_ex(_p(xd), ci, _p(wd), _p(bd), None, _p(y), None, 0, co // 8, 0, 1, 1, n, h, w, co, stride,
                                                0.01, S), "enc_conv")
        out = _f8_decode(y, co).cpu()
        tol = 4e-5 + 2e-5 * float(ref.abs().max())            # (+ the 15-bit read-back of the f8 container itself)
    else:
        y = torch.empty([n, co, ho, wo], dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().nb_enc_conv3x3_ex(_p(xd), ci, _p(wd), _p(bd), _p(y), None, None, 0, 0, 0, 1, 0, n, h, w, co, stride, 0.01, S),
                   "enc_conv")
        out = y.cpu()
        tol = 4e-5 * float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) <= tol, (float((out.double() - ref).abs().max()), tol)


@pytest.mark.parametrize("res,n,pre", [(128, 9, None), (256, 8, "-11inverse"), (384, 8, None)])
def test_encoder_f8_matches_oracle(res, n, pre):
    """Batches >= f8_min_batch run the encoder with f8 operands between its layers: features within 5e-4 of the fp32 oracle
    (O(1) values; 2e-4 for the hi/lo-f16 path), and the two formats agree to the same level."""
    rs = np.random.RandomState(res + n)
    esd = encmod.random_encoder_state_dict(11)
    geom = (rs.rand(n, 1, res, res) > 0.1).astype(np.float32)
    geom[:, :, ::7, :] = rs.rand(n, 1, len(range(0, res, 7)), res).astype(np.float32)
    enc = encmod.HipGeometryEncoder(esd, preproc_type=pre)
    assert enc.arith == "f8" and n >= enc.f8_min_batch
    f = enc.encode(torch.from_numpy(geom).cuda())
    ref = po.encoder_encode(esd, torch.from_numpy(geom), pre)
    enc.arith = "h3"
    g = enc.encode(torch.from_numpy(geom).cuda())
    for a, b, c in zip(f, ref, g):
        assert float((a.cpu() - b).abs().max()) <= 5e-4, float((a.cpu() - b).abs().max())
        assert float((c.cpu() - b).abs().max()) <= TOL
        assert not torch.equal(a, c)                              # (the f8 path really ran)


@pytest.mark.parametrize("n,h,w,co,pre,out_fmt", [(2, 64, 64, 128, 0, 1), (1, 32, 128, 128, 1, 1), (2, 16, 64, 48, 2, 0), (1, 128, 192, 144, 0, 1),
                                                  (3, 256, 256, 128, 0, 1)])
def test_enc_fused_stem_conv(n, h, w, co, pre, out_fmt):
    """Stem + first stride-2 stage in one launch (nb_enc_stem_conv3x3_f8: the stem's outputs are computed per 16-channel chunk on the
    matrix pipe, straight into the slab buffers) against torch fp64 of both layers and against the two-kernel path it replaces
    (nb_enc_stem7x7_f32_h2_ex -> nb_enc_conv3x3_ex): one tile and many, the first / last tile rows and columns (both reflect
    paddings), ragged c_out slices, every preprocessing type, both output formats."""
    rs = np.random.RandomState(n + h + w + co)
    x = torch.from_numpy(rs.rand(n, 1, h, w).astype(np.float32))
    w0 = (rs.randn(64, 1, 7, 7) / 7).astype(np.float32)
    b0 = rs.randn(64).astype(np.float32)
    w1 = (rs.randn(co, 64, 3, 3) / np.sqrt(9 * 64)).astype(np.float32)
    b1 = rs.randn(co).astype(np.float32)
    xp = x.double()
    xp = (1 - xp) * 2 - 1 if pre == 1 else (1 - xp if pre == 2 else xp)
    F = torch.nn.functional
    s = F.leaky_relu(F.conv2d(F.pad(xp, (3, 3, 3, 3), mode="reflect"), torch.from_numpy(w0).double(), torch.from_numpy(b0).double()), 0.01)
    ref = F.leaky_relu(F.conv2d(F.pad(s, (1, 1, 1, 1), mode="reflect"), torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), stride=2), 0.01)
    w50 = np.zeros([64, 50], np.float32)
    w50[:, :49] = w0.reshape(64, 49)
    xd, w50d, b0d = x.cuda(), torch.from_numpy(w50).cuda(), torch.from_numpy(b0).cuda()
    w1d, b1d = torch.from_numpy(encmod.pack_enc_weight_f8(w1)).cuda(), torch.from_numpy(b1).cuda()
    S = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    y = torch.zeros([n, co // 8, 2, h // 2, w // 2, 8], dtype=torch.float16, device="cuda")
    _lib.check(lib.nb_enc_stem_conv3x3_f8(_p(xd), _p(w50d), _p(b0d), pre, _p(w1d), _p(b1d), _p(y), out_fmt, n, h, w, co, 0.01, S), "fused")
    dec = (lambda t: _f8_decode(t, co)) if out_fmt else _h2_to_nchw
    out = dec(y).cpu().double()
    tol = 4e-5 + 2e-5 * float(ref.abs().max())
    assert float((out - ref).abs().max()) <= tol, (float((out - ref).abs().max()), tol)
    # the two-kernel path
    a = torch.empty([n, 8, 2, h, w, 8], dtype=torch.float16, device="cuda")
    y2 = torch.zeros_like(y)
    _lib.check(lib.nb_enc_stem7x7_f32_h2_ex(_p(xd), _p(w50d), _p(b0d), _p(a), 1, n, h, w, pre, 0.01, S), "stem")
    _lib.check(lib.nb_enc_conv3x3_ex(_p(a), 64, _p(w1d), _p(b1d), None, _p(y2), None, 0, co // 8, 0, 1, out_fmt, n, h, w, co, 2, 0.01, S), "conv")
    d = float((dec(y2).cpu().double() - out).abs().max())
    assert d <= 2 * tol, (d, tol)                                 # (each within tol of the truth; not bit-identical: another summation order in the stem)


@pytest.mark.parametrize("res,n,pre", [(128, 8, "inverse"), (256, 9, None)])
def test_encoder_fused_stem_equals_two_launches(res, n, pre):
    """HipGeometryEncoder with the stem inside the first stride-2 launch (the default for f8 batches) against the same encoder with the two
    launches (fuse_stem = False): both features agree to the f8 path's own tolerance against the oracle, and they are not the same bits (the
    fused launch really ran: another summation order in the stem)."""
    rs = np.random.RandomState(res + n)
    esd = encmod.random_encoder_state_dict(13)
    geom = torch.from_numpy((rs.rand(n, 1, res, res) ** 3).astype(np.float32)).cuda()
    enc = encmod.HipGeometryEncoder(esd, preproc_type=pre)
    assert enc.fuse_stem and enc.arith == "f8"
    a = enc.encode(geom)
    enc.fuse_stem = False
    b = enc.encode(geom)
    ref = po.encoder_encode(esd, geom.cpu(), pre)
    for x, y, r in zip(a, b, ref):
        assert float((x.cpu() - r).abs().max()) <= 5e-4 and float((y.cpu() - r).abs().max()) <= 5e-4
        assert float((x - y).abs().max()) <= 2e-4
    assert not all(torch.equal(x, y) for x, y in zip(a, b))
