"""GPU parity of the split-f16 ("h3") fast path: H2 packing, the f16-MFMA conv1 kernel and the fp32 up=2
kernel's H2 output mode, each against the CPU oracle (fp32).  Tolerance 5e-5 on O(1) activations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def D(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def maxerr(got, want):
    got = got.detach().cpu().numpy().astype(np.float64)
    want = (want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)).astype(np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.abs(got - want).max())


def test_pack_unpack_h2(dev):
    from brushstroke_engine_amd import ops
    rs = np.random.RandomState(0)
    x = (rs.randn(2, 20, 16, 32) * 3).astype(np.float32)
    x2 = rs.randn(2, 5, 16, 32).astype(np.float32)
    s = (1 + 0.3 * rs.randn(2, 25)).astype(np.float32)
    h2 = ops.pack_h2(D(x, dev), D(s, dev), D(x2, dev))
    assert list(h2.shape) == [2, 4, 2, 16, 32, 8]
    want = np.concatenate([x, x2], 1) * s[:, :, None, None]
    got = ops.unpack_h2(h2, 25)
    assert maxerr(got, want) <= 2e-6 * np.abs(want).max()
    assert float(h2[:, 3, :, :, :, 1:].abs().max()) == 0.0          # channel padding is zero


@pytest.mark.parametrize("shape", [(2, 128, 128, 32, 64), (1, 64, 64, 64, 32), (3, 40, 96, 16, 32), (1, 128, 128, 128, 128)])
def test_up1_h3_vs_oracle(dev, shape):
    from brushstroke_engine_amd import ops
    from oracle import neube_oracle as orc
    n, ic, oc, h, w = shape
    rs = np.random.RandomState(ic + oc + h)
    x = rs.randn(n, ic, h, w).astype(np.float32)
    wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, h, w)).astype(np.float32)
    T = torch.from_numpy
    want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=1, padding=1, flip_weight=True)
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0)
    wd = D(wt, dev)
    w_h3 = ops.pack_conv_weight_h3(wd)
    xh2 = ops.pack_h2(D(x, dev), D(s, dev))
    wsq = wd.square().sum(dim=[2, 3]).t().contiguous()
    d = (D(s, dev).square() @ wsq + 1e-8).rsqrt()
    got = ops.modconv_up1_h3(xh2, ic, w_h3, d, D(noise, dev), D(b, dev), oc, act_clamp=256.0)
    assert maxerr(got, want) <= 5e-5


@pytest.mark.parametrize("shape", [(2, 144, 128, 32), (1, 128, 64, 64), (2, 36, 32, 16)])
def test_up2_h2_output_vs_oracle(dev, shape):
    from brushstroke_engine_amd import ops, _lib
    from oracle import neube_oracle as orc
    n, ic, oc, h = shape
    rs = np.random.RandomState(ic + oc + h)
    x = rs.randn(n, ic, h, h).astype(np.float32)
    wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    s_next = (1 + 0.5 * rs.randn(n, oc)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, 2 * h, 2 * h)).astype(np.float32)
    T = torch.from_numpy
    want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=2, padding=1, resample_filter=orc.setup_filter(),
                                flip_weight=False)
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0) * T(s_next)[:, :, None, None]
    wd, sd_ = D(wt, dev), D(s, dev)
    wpk, wsq = ops.pack_conv_weight(wd)
    d = (sd_.square() @ wsq + 1e-8).rsqrt()
    out = torch.zeros(ops.h2_shape(n, oc, 2 * h, 2 * h), dtype=torch.float16, device=dev)
    xd, nd, bd, snd = D(x, dev), D(noise, dev), D(b, dev), D(s_next, dev)
    rc = _lib.lib().nb_modconv3x3_up2_f32_h2(xd.data_ptr(), ic, None, 0, wpk.data_ptr(), sd_.data_ptr(), d.data_ptr(),
                                             nd.data_ptr(), 4 * h * h if n > 1 else 0, bd.data_ptr(), snd.data_ptr(),
                                             out.data_ptr(), n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0,
                                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "up2_h2")
    got = ops.unpack_h2(out, oc)
    assert maxerr(got, want) <= 5e-5


@pytest.mark.parametrize("shape", [(2, 144, 128, 32, 32), (1, 384, 128, 64, 64), (1, 128, 64, 128, 128), (2, 40, 48, 16, 32),
                                   (3, 128, 128, 16, 16), (2, 48, 40, 24, 16)])
def test_up2_h3_vs_oracle(dev, shape):
    from brushstroke_engine_amd import ops
    from oracle import neube_oracle as orc
    n, ic, oc, h, w = shape
    rs = np.random.RandomState(ic + oc + h)
    x = rs.randn(n, ic, h, w).astype(np.float32)
    wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, 2 * h, 2 * w)).astype(np.float32)
    T = torch.from_numpy
    want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=2, padding=1, resample_filter=orc.setup_filter(),
                                flip_weight=False)
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0)
    wd = D(wt, dev)
    c2 = 16 if ic > 64 else 0
    xh2 = ops.pack_h2(D(x[:, :ic - c2], dev), D(s, dev), D(x[:, ic - c2:], dev) if c2 else None)
    wsq = wd.square().sum(dim=[2, 3]).t().contiguous()
    d = (D(s, dev).square() @ wsq + 1e-8).rsqrt()
    got = ops.modconv_up2_h3(xh2, ic, ops.pack_conv_weight_h3(wd), d, D(noise, dev), D(b, dev), oc, act_clamp=256.0)
    assert maxerr(got, want) <= 5e-5


# ---------------------------------------------------------------- fused H2 hand-off between split-f16 layers
@pytest.mark.parametrize("up,ci,co,res,c_next", [(1, 64, 64, 64, 64), (1, 128, 128, 32, 384), (2, 128, 64, 64, 64),
                                                  (2, 144, 128, 64, 128), (2, 128, 128, 32, 128)])
def test_h2_handoff_kernels(up, ci, co, res, c_next):
    """nb_modconv3x3_up{1,2}_h3_h2 == the fp32-output kernel followed by nb_pack_h2_f32 with the consumer's styles."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(ci + co + up)
    n = 3
    hin = res if up == 1 else res // 2
    x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, c_next)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
    xh, wp = ops.pack_h2(x, st), ops.pack_conv_weight_h3(w)
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    y = torch.empty([n, co, res, res], device="cuda")
    f32 = lib.nb_modconv3x3_up1_h3 if up == 1 else lib.nb_modconv3x3_up2_h3
    h2 = lib.nb_modconv3x3_up1_h3_h2 if up == 1 else lib.nb_modconv3x3_up2_h3_h2
    _lib.check(f32(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(), y.data_ptr(),
                   n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f32")
    out = torch.zeros(ops.h2_shape(n, c_next, res, res), dtype=torch.float16, device="cuda")
    _lib.check(h2(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(), nst.data_ptr(),
                  c_next, out.data_ptr(), c_next, n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "h2")
    ref = ops.pack_h2(y, nst[:, :co].contiguous())
    # same value; the hi/lo split may differ where y*style is an exact f16 tie (the fused kernel rounds the product
    # once, the two-pass path twice), so compare hi+lo, which carries 22 bits either way
    got, want = ops.unpack_h2(out[:, :co // 8].contiguous(), co), ops.unpack_h2(ref, co)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert (out[:, :co // 8] != ref).float().mean() < 1e-3
    assert not out[:, co // 8:].any()                       # channel groups of the geometry features are left alone
    if c_next > co:
        g = torch.from_numpy(rs.randn(n, c_next - co, res, res).astype(np.float32)).cuda()
        _lib.check(lib.nb_pack_h2_part_f32(g.data_ptr(), c_next - co, nst.data_ptr() + 4 * co, c_next, out.data_ptr(),
                                           (c_next + 7) // 8, co // 8, n, res * res, S), "part")
        full = ops.pack_h2(y, nst, g)
        assert torch.equal(out[:, co // 8:], full[:, co // 8:])


def test_generator_h2_handoff_equals_pack_path():
    """End to end: the fused hand-off changes no bit of the output."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod
    from brushstroke_engine_amd.networks import Generator
    cfg = cfgmod.style1_config(256)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode="h3").to("cuda")
    from brushstroke_engine_amd import synthetic
    n = 4
    z = torch.from_numpy(synthetic.batch_z(cfg, n)).cuda()
    gf = [torch.from_numpy(a).cuda() for a in synthetic.geom_features(cfg, n, seed=1)]
    pos = torch.from_numpy(synthetic.positions(cfg, n, seed=1)).cuda()
    G.synthesis.h2_handoff = True
    a, _, da = G.render_triad(z=z, geom_feature=gf, positions=pos)
    G.synthesis.h2_handoff = False
    b, _, db = G.render_triad(z=z, geom_feature=gf, positions=pos)
    assert float((da["uvs"] - db["uvs"]).abs().max()) < 2e-6        # 22-bit hand-off either way
    assert int((a.int() - b.int()).abs().max()) <= 1 and float((a != b).float().mean()) < 1e-4
    # taps still work with the hand-off enabled (fp32 features at the blending resolution)
    G.synthesis.h2_handoff = True
    _, _, dc = G.render_triad(z=z, geom_feature=gf, positions=pos, return_features=[128])
    assert float((dc["uvs"] - da["uvs"]).abs().max()) < 2e-6 and dc["features128"].shape == (n, 128, 128, 128)


@pytest.mark.parametrize("res", [128, 256])
def test_fused_torgb_equals_standalone(res):
    """Last conv1 + ToRGB + compositing in one launch == the two-launch form (same arithmetic order), all outputs."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.networks import Generator
    cfg = cfgmod.style1_config(res)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode="h3").to("cuda")
    n = 5
    z = torch.from_numpy(synthetic.batch_z(cfg, n)).cuda()
    gf = [torch.from_numpy(a).cuda() for a in synthetic.geom_features(cfg, n, seed=1)]
    pos = torch.from_numpy(synthetic.positions(cfg, n, seed=1)).cuda()
    user = torch.full([n, 3, 3], float("nan")); user[1, :, 0] = torch.tensor([1.0, 0.0, 0.25])
    kw = dict(z=z, geom_feature=gf, positions=pos, want_f32=True, user_colors=user, sfactor=torch.tensor(1.3), render_mode="clear")
    outs = {}
    for fused in (True, False):
        G.synthesis.fuse_torgb = fused
        u8, f32, dbg = G.render_triad(**kw)
        img, dbg2 = G(z, None, gf, positions=pos, noise_mode="const", return_debug_data=True, return_features=[res])
        outs[fused] = (u8, f32, dbg["uvs"], dbg["colors"], img, dbg2[f"features{res}"])
    for a, b in zip(outs[True][1:], outs[False][1:]):
        assert torch.equal(a, b)
    assert torch.equal(outs[True][0], outs[False][0])


# ---------------------------------------------------------------- small-image split-f16 conv1 kernel
@pytest.mark.parametrize("shape", [(1, 128, 128, 4), (3, 128, 128, 4), (2, 128, 128, 8), (1, 128, 96, 16), (2, 64, 128, 32),
                                   (1, 128, 128, 64), (5, 48, 40, 8), (2, 256, 64, 16)])
@pytest.mark.parametrize("waves", [0, 4, 8])
def test_up1_small_h3_vs_oracle(dev, shape, waves):
    """nb_modconv3x3_up1_small_h3 (32 x 32 tiles, K split over 4 or 8 waves -- 0 = the library's rule; 4x4 images two samples
    per tile, odd batch = half-empty last tile) against the fp32 CPU oracle.  Tolerance 5e-5 on O(1) activations, like the
    large-tile kernel."""
    from brushstroke_engine_amd import ops, _lib
    from oracle import neube_oracle as orc
    n, ic, oc, h = shape
    rs = np.random.RandomState(ic + oc + h + n)
    x = rs.randn(n, ic, h, h).astype(np.float32)
    wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, h, h)).astype(np.float32)
    T = torch.from_numpy
    want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=1, padding=1, flip_weight=True)
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0)
    wd, sd_ = D(wt, dev), D(s, dev)
    wsq = wd.square().sum(dim=[2, 3]).t().contiguous()
    d = (sd_.square() @ wsq + 1e-8).rsqrt()
    w_h3 = ops.pack_conv_weight_h3(wd)
    xd, nd, bd = D(x, dev), D(noise, dev), D(b, dev)
    y = torch.full([n, oc, h, h], float("nan"), dtype=torch.float32, device=dev)
    _lib.lib().nb_debug_set_small_waves(waves)
    try:
        rc = _lib.lib().nb_modconv3x3_up1_small_h3(xd.data_ptr(), ic, w_h3.data_ptr(), sd_.data_ptr(), d.data_ptr(), nd.data_ptr(),
                                                   h * h, bd.data_ptr(), y.data_ptr(), n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0,
                                                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        _lib.lib().nb_debug_set_small_waves(0)
    _lib.check(rc, "small_h3")
    assert maxerr(y, want) <= 5e-5


@pytest.mark.parametrize("up", [1, 2])
@pytest.mark.parametrize("n,ic,oc,h", [(2, 128, 128, 16), (1, 128, 96, 64), (3, 128, 128, 8), (2, 256, 64, 32), (5, 144, 128, 32)])
def test_small_h3_two_block_tiles_bit_identical(dev, up, n, ic, oc, h):
    """The small-image kernel's two-block tiles (64 positions per workgroup: the weight fragments serve both blocks; taken when a
    launch is more than a round of workgroups) compute every position exactly as the one-block tiles do."""
    from brushstroke_engine_amd import ops, _lib
    from oracle import neube_oracle as orc
    rs = np.random.RandomState(ic + oc + h + n + up)
    x = D(rs.randn(n, ic, h, h).astype(np.float32), dev)
    wt = D(rs.randn(oc, ic, 3, 3).astype(np.float32), dev)
    sd_ = D((1 + 0.5 * rs.randn(n, ic)).astype(np.float32), dev)
    bd = D((0.1 * rs.randn(oc)).astype(np.float32), dev)
    nd = D((0.1 * rs.randn(n, 1, up * h, up * h)).astype(np.float32), dev)
    d = (sd_.square() @ wt.square().sum(dim=[2, 3]).t().contiguous() + 1e-8).rsqrt()
    wp = ops.pack_conv_weight_h3(wt) if up == 1 else ops.pack_conv_weight_h3_up2_phases(wt, orc.setup_filter().to(dev))
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for blocks in (1, 2):
            lib.nb_debug_set_small_blocks(blocks)
            y = torch.full([n, oc, up * h, up * h], float("nan"), dtype=torch.float32, device=dev)
            if up == 1:
                rc = lib.nb_modconv3x3_up1_small_h3(x.data_ptr(), ic, wp.data_ptr(), sd_.data_ptr(), d.data_ptr(), nd.data_ptr(), h * h,
                                                    bd.data_ptr(), y.data_ptr(), n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0, S)
            else:
                rc = lib.nb_modconv3x3_up2_small_h3(x.data_ptr(), ic, None, 0, wp.data_ptr(), sd_.data_ptr(), d.data_ptr(), nd.data_ptr(),
                                                    4 * h * h, bd.data_ptr(), y.data_ptr(), n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0, S)
            torch.cuda.synchronize()
            _lib.check(rc, "small_h3")
            out[blocks] = y
    finally:
        lib.nb_debug_set_small_blocks(0)
    assert not torch.isnan(out[2]).any()
    assert torch.equal(out[1], out[2])


@pytest.mark.parametrize("shape", [(1, 128, 0, 128, 4), (3, 128, 0, 128, 4), (2, 128, 0, 128, 8), (1, 128, 0, 96, 16),
                                   (1, 128, 16, 128, 32), (2, 32, 16, 40, 8)])
@pytest.mark.parametrize("waves", [0, 8])
def test_up2_small_h3_vs_oracle(dev, shape, waves):
    """nb_modconv3x3_up2_small_h3 (FIR folded into four per-phase 3x3 kernels, phases as grid.z; optional concatenated
    second input) against the fp32 CPU oracle's conv_transpose2d + upfirdn2d.  Tolerance 5e-5 on O(1) activations."""
    from brushstroke_engine_amd import ops, _lib
    from oracle import neube_oracle as orc
    n, c1, c2, oc, h = shape
    ic = c1 + c2
    rs = np.random.RandomState(ic + oc + h + n)
    x = rs.randn(n, ic, h, h).astype(np.float32)
    wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, 2 * h, 2 * h)).astype(np.float32)
    T = torch.from_numpy
    f = orc.setup_filter()
    want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=2, padding=1, resample_filter=f, flip_weight=False)
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0)
    wd, sd_ = D(wt, dev), D(s, dev)
    wsq = wd.square().sum(dim=[2, 3]).t().contiguous()
    d = (sd_.square() @ wsq + 1e-8).rsqrt()
    wph = ops.pack_conv_weight_h3_up2_phases(wd, f.to(dev))
    x1 = D(x[:, :c1], dev)
    x2 = D(x[:, c1:], dev) if c2 else None
    nd, bd = D(noise, dev), D(b, dev)
    y = torch.full([n, oc, 2 * h, 2 * h], float("nan"), dtype=torch.float32, device=dev)
    _lib.lib().nb_debug_set_small_waves(waves)
    try:
        rc = _lib.lib().nb_modconv3x3_up2_small_h3(x1.data_ptr(), c1, None if x2 is None else x2.data_ptr(), c2, wph.data_ptr(),
                                                   sd_.data_ptr(), d.data_ptr(), nd.data_ptr(), 4 * h * h, bd.data_ptr(), y.data_ptr(),
                                                   n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        _lib.lib().nb_debug_set_small_waves(0)
    _lib.check(rc, "small_h3_up2")
    assert maxerr(y, want) <= 5e-5
