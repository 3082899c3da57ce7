"""GPU parity of the "f8" conv mode (correction products of the split-f16 scheme on block-scaled fp8 MFMAs):
kernels against float64 / the h3 kernels, the generator against the REFERENCE outputs (tests/golden/gen_r*.npz) and the
tiled canvas against the reference-painted canvas.  north_star tolerance: 1e-3 max abs on pixels; asserted: 3e-4."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
PIX_TOL_F8 = 3e-4


def _conv_ref(x, w, st, up):
    xm = (x * st[:, :, None, None]).double().cpu()
    if up == 1:
        return torch.nn.functional.conv2d(xm, w.double().cpu(), padding=1)
    from oracle import neube_oracle as orc
    f = orc.setup_filter([1, 3, 3, 1], dtype=torch.float64)
    return orc.conv2d_resample(xm, w.double().cpu(), f=f, up=2, padding=1, flip_weight=False)


@pytest.mark.parametrize("up,ci,co,res", [(1, 64, 64, 64), (1, 128, 128, 32), (1, 144, 96, 64), (2, 128, 64, 64),
                                           (2, 384, 128, 64), (2, 144, 128, 64), (2, 128, 128, 32)])
def test_f8_kernels_vs_float64(up, ci, co, res):
    """One layer, f8 operands, fp32 output: error against float64 at the level of the fp8 correction terms
    (2^-11 * 2^-4 relative per product), ~20x the h3 error and ~30x below a plain f16 evaluation."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(up * 1000 + ci + co)
    n = 2
    hin = res if up == 1 else res // 2
    x = torch.from_numpy((rs.randn(n, ci, hin, hin) * 2).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    dco, bias = torch.ones(n, co, device="cuda"), torch.zeros(co, device="cuda")
    ref = _conv_ref(x, w, st, up)
    S = torch.cuda.current_stream().cuda_stream
    errs = {}
    for fmt, pack_x, pack_w in ((0, ops.pack_h2, ops.pack_conv_weight_h3), (1, ops.pack_h2f8, ops.pack_conv_weight_h3f8)):
        xh, wp = pack_x(x, st), pack_w(w)
        y = torch.empty([n, co, res, res], device="cuda")
        if up == 1:
            rc = _lib.lib().nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(),
                                                    y.data_ptr(), None, None, 0, 0, None, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
        else:
            rc = _lib.lib().nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(),
                                                    y.data_ptr(), None, None, 0, 0, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
        _lib.check(rc, "conv")
        errs[fmt] = float((y.cpu().double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert errs[0] <= 2e-6 * scale and errs[1] <= 4e-5 * scale, (errs, scale)


@pytest.mark.parametrize("up,ci,co,res,c_next", [(1, 64, 64, 64, 64), (1, 128, 128, 32, 384), (2, 128, 64, 64, 64), (2, 144, 128, 64, 128),
                                                  (2, 128, 128, 32, 128)])
def test_f8_handoff_equals_pack(up, ci, co, res, c_next):
    """f8-format output of a producer == fp32 output followed by nb_pack_h2f8_f32 with the consumer's styles
    (same value; the fp8 bytes may differ where the product is an exact tie, so the decoded planes are compared)."""
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
    xh, wp = ops.pack_h2f8(x, st), ops.pack_conv_weight_h3f8(w)
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    y = torch.empty([n, co, res, res], device="cuda")
    out = torch.zeros(ops.h2_shape(n, c_next, res, res), dtype=torch.float16, device="cuda")
    common = (dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr())
    if up == 1:
        _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, None, 1, 0,
                                               n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f32")
        _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), c_next,
                                               c_next, None, 1, 1, n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f8out")
    else:
        _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, 1, 0,
                                               n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f32")
        _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), c_next,
                                               c_next, 1, 1, n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f8out")
    ref = ops.pack_h2f8(y, nst[:, :co].contiguous())

    def decode(t, c):
        """f8-format tensor -> (hi f32 [n,c,h,w], xl8 f32, xh8 f32)"""
        nn, c8, _, h, w_, _ = t.shape
        hi = t[:, :, 0].float().permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
        lo = t[:, :, 1].contiguous().view(torch.uint8).view(torch.float8_e4m3fn).float()      # [n, c8, h, w, 16]
        xl = lo[:, 0::2].permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
        xh = lo[:, 1::2].permute(0, 1, 4, 2, 3).reshape(nn, c8 * 8, h, w_)[:, :c]
        return hi, xl, xh
    got, want = decode(out[:, :co // 8].contiguous(), co), decode(ref, co)
    v_got = got[0] + got[1] / 512
    v_want = want[0] + want[1] / 512
    assert float((v_got - v_want).abs().max()) <= 2e-5 * float(v_want.abs().max())        # hi + xl: ~15 bits either way
    assert float((got[2] - want[2]).abs().max()) <= 0.07 * float(want[2].abs().max())      # fp8(v/4): one fp8 step at most
    assert (out[:, :co // 8] != ref).float().mean() < 2e-3
    assert not out[:, co // 8:].any()
    if c_next > co:
        g = torch.from_numpy(rs.randn(n, c_next - co, res, res).astype(np.float32)).cuda()
        _lib.check(lib.nb_pack_h2f8_part_f32(g.data_ptr(), c_next - co, nst.data_ptr() + 4 * co, c_next, out.data_ptr(),
                                             c_next // 8, co // 8, n, res * res, S), "part")
        full = ops.pack_h2f8(y, nst, g)
        assert torch.equal(out[:, co // 8:], full[:, co // 8:])


@pytest.mark.parametrize("res", [128, 256])
def test_f8_generator_vs_reference_golden(res):
    """style1 shapes against the reference's outputs in the f8 mode: pixels within 3e-4 (budget 1e-3)."""
    from brushstroke_engine_amd import config as cfgmod, synthetic, weights as wmod
    from brushstroke_engine_amd.networks import Generator
    g = load_golden(f"gen_r{res}.npz")
    cfg = cfgmod.style1_config(res)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=int(g["weights_seed"])), conv_mode="f8").to("cuda")
    geom = [torch.from_numpy(x).cuda() for x in synthetic.geom_features(cfg, 2, seed=int(g["geom_seed"]))]
    z = torch.from_numpy(np.concatenate([g["z"]] * 2)).cuda()              # batch 4 so that the split-f16 path is taken
    geom = [torch.cat([x, x]) for x in geom]
    pos = torch.from_numpy(np.concatenate([g["positions"]] * 2)).cuda()
    img, dbg = G(z, None, geom, positions=pos, return_debug_data=True, noise_mode="const")
    assert any("_h3" in k for k in G.synthesis.layer_kernels.values())
    step = int(g["step"])
    for name, full in (("uvs", dbg["uvs"]), ("img", img)):
        got = full[:2].cpu().numpy()[..., ::step, ::step]
        assert float(np.abs(got - g[f"{name}.sub"]).max()) <= PIX_TOL_F8, name
    assert float(np.abs(dbg["uvs"][:2, :, res // 3, :].cpu().numpy() - g["uvs.row"]).max()) <= PIX_TOL_F8
    assert torch.equal(img[:2], img[2:])                                    # batch-composition independence, bitwise


def test_f8_tiled_canvas_matches_reference():
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
    from brushstroke_engine_amd.networks import Generator
    g = load_golden("engine_r128.npz")
    cfg = cfgmod.style1_config(128)
    G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode="f8").to("cuda")
    ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5)))
    helper = painting.PaintingHelper(ops, batch=4)
    helper.set_feature_blending(2)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
    _, full, _, _ = helper.paint_image(g["geom"], opts, crop_margin=int(g["crop_margin"]), return_full=True)
    d = np.abs(full.astype(np.int32) - g["canvas_level2_clear"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 5e-3, (d.max(), (d > 0).mean())


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("ci,co,h,w", [(128, 64, 32, 32), (144, 128, 24, 64), (48, 64, 26, 32), (128, 64, 8, 8), (32, 32, 16, 16)])
def test_up2_tile_heights_agree(fmt, ci, co, h, w):
    """The up=2 split-f16 kernel has three tile heights (12 quad rows for throughput, 8 where those would end in a mostly
    empty round of workgroups, 5 for under-filled launches such as batch 1).  Both walk the same per-pixel arithmetic, so fp32 and hand-off outputs must be bit-identical; 24 rows do
    not divide by 5 (overhanging last tile)."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(ci + h + fmt)
    n = 2
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, 2 * h, 2 * w).astype(np.float32)).cuda()
    xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
    wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(wt)
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for tqh in (12, 8, 5, "pair", "v2"):
            lib.nb_debug_set_up2_tile(12 if tqh == "pair" else 0 if tqh == "v2" else tqh)
            lib.nb_debug_set_up2_pair(1 if tqh == "pair" else 0)
            lib.nb_debug_set_up2_v2(1 if tqh == "v2" else 0)        # (12 = the round-3 kernel on the same tiles)
            y = torch.empty([n, co, 2 * h, 2 * w], device="cuda")
            out = torch.zeros(ops.h2_shape(n, co, 2 * h, 2 * w), dtype=torch.float16, device="cuda")
            common = (dco.data_ptr(), noise.data_ptr(), 4 * h * w, bias.data_ptr())
            _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, fmt, 0,
                                                   n, h, w, co, 0.2, 1.4142135, 256.0, S), "f32 out")
            _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), co,
                                                   co, fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S), "hand-off out")
            torch.cuda.synchronize()
            res[tqh] = (y, out)
    finally:
        lib.nb_debug_set_up2_tile(0)
        lib.nb_debug_set_up2_pair(-1)
        lib.nb_debug_set_up2_v2(-1)
    # the 8-wave kernel with the software-pipelined K loop (same tiles as 12; f8 and H2 operands)
    assert torch.equal(res[12][0], res["v2"][0])
    assert torch.equal(res[12][1], res["v2"][1])
    assert torch.equal(res[12][0], res[5][0])
    assert torch.equal(res[12][1], res[5][1])
    assert torch.equal(res[12][0], res[8][0])                # 8-row tiles (launches that would end in a mostly empty round of 12-row tiles)
    assert torch.equal(res[12][1], res[8][1])
    assert torch.equal(res[12][0], res["pair"][0])           # two 4-wave workgroups per CU on 12 x 16 tiles: the same arithmetic
    assert torch.equal(res[12][1], res["pair"][1])
    # and against float64 (loose: the exact bounds live in test_f8_kernels_vs_float64)
    ref = _conv_ref(x, wt, st, 2) * dco.double().cpu()[:, :, None, None] + noise.double().cpu()[:, None]
    ref = torch.nn.functional.leaky_relu(ref + bias.double().cpu()[None, :, None, None], 0.2) * 1.4142135
    assert float((res[5][0].double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("ci,co,h,w", [(128, 128, 32, 64), (64, 64, 48, 32), (48, 64, 32, 32), (16, 128, 16, 32), (40, 64, 32, 32)])
def test_up1_rows_per_wave_agree(fmt, ci, co, h, w):
    """The 8-wave up=1 split-f16 kernel runs with 2 (throughput) or 1 (under-filled launches, batch 1) pixel rows per
    wave; same per-pixel arithmetic, so fp32 and hand-off outputs must be bit-identical."""
    from brushstroke_engine_amd import _lib, ops
    if fmt and ci % 16:
        pytest.skip("f8 operands need whole 16-channel chunks (an odd number of channel groups is an H2 case: the last chunk's missing group "
                    "comes from the zero page, and the software-pipelined loop leaves such layers to the round-3 loop)")
    rs = np.random.RandomState(ci + h + fmt)
    n = 2
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, h, w).astype(np.float32)).cuda()
    xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
    wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(wt)
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for rows in (2, 1, "2old", "1old") + (("2pp",) if fmt else ()):
            # ("old": the round-3 K loop; the default is the software-pipelined one, in both operand formats; "pp": round 5's ping-pong
            #  form of the f8 loop -- waves 4-7 one segment behind waves 0-3, loads against matrix work)
            lib.nb_debug_set_up1_rows(int(str(rows)[0]))
            lib.nb_debug_set_up1_v2(0 if str(rows).endswith("old") else 1)
            lib.nb_debug_set_up1_pp(1 if str(rows).endswith("pp") else 0)
            y = torch.empty([n, co, h, w], device="cuda")
            out = torch.zeros(ops.h2_shape(n, co, h, w), dtype=torch.float16, device="cuda")
            common = (dco.data_ptr(), noise.data_ptr(), h * w, bias.data_ptr())
            _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, None, fmt, 0,
                                                   n, h, w, co, 0.2, 1.4142135, 256.0, S), "f32 out")
            _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), co,
                                                   co, None, fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S), "hand-off out")
            torch.cuda.synchronize()
            res[rows] = (y, out)
    finally:
        lib.nb_debug_set_up1_rows(0)
        lib.nb_debug_set_up1_v2(-1)
        lib.nb_debug_set_up1_pp(-1)
    assert torch.equal(res[2][0], res[1][0])
    assert torch.equal(res[2][1], res[1][1])
    for k in ("2old", "1old") + (("2pp",) if fmt else ()):
        assert torch.equal(res[2][0], res[k][0]) and torch.equal(res[2][1], res[k][1]), k
    ref = _conv_ref(x, wt, st, 1) * dco.double().cpu()[:, :, None, None] + noise.double().cpu()[:, None]
    ref = torch.nn.functional.leaky_relu(ref + bias.double().cpu()[None, :, None, None], 0.2) * 1.4142135
    assert float((res[1][0].double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_generator_batch1_tiles_equal_forced_large_tiles():
    """Whole generator at batch 1 (small tiles chosen automatically, fused ToRGB included) == the same with the
    throughput tiles forced: bit-identical RGBA."""
    from brushstroke_engine_amd import _lib, config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.networks import Generator
    dev = torch.device("cuda:0")
    cfg = cfgmod.style1_config(256)
    G = Generator(cfg, wmod.random_state_dict(cfg, 3), conv_mode="f8").to(dev)
    z = torch.from_numpy(synthetic.batch_z(cfg, 1, 11)).to(dev)
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, 1, 5)]
    pos = torch.from_numpy(synthetic.positions(cfg, 1, 5)).to(dev)
    lib = _lib.lib()
    a = G.render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
    try:
        lib.nb_debug_set_up1_rows(2); lib.nb_debug_set_up2_tile(12)
        b = G.render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
        torch.cuda.synchronize()
    finally:
        lib.nb_debug_set_up1_rows(0); lib.nb_debug_set_up2_tile(0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2]["uvs"], b[2]["uvs"])


@pytest.mark.parametrize("up,ci,co,res,n", [(1, 128, 128, 64, 16), (1, 64, 64, 128, 8), (2, 128, 64, 128, 8), (2, 384, 128, 64, 32)])
def test_hi_only_form_is_the_plain_f16_product(up, ci, co, res, n):
    """in_fmt 3 (Generator(conv_mode="f16"), round 6): f8 operands and weights with the correction products skipped on the large
    throughput kernels -- the output must be the convolution of the f16-ROUNDED operands (fp32 accumulation: ~1e-6 of the largest
    output from a float64 evaluation of those rounded operands), i.e. a plain single-f16 evaluation, ~1e-3 relative away from the
    f8 result (whose corrections restore the fp32 operands to ~2^-15)."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(up * 100 + ci + res)
    hin = res if up == 1 else res // 2
    lib = _lib.lib()
    x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    dco, bias = torch.ones(n, co, device="cuda"), torch.zeros(co, device="cuda")
    S = torch.cuda.current_stream().cuda_stream
    xh, wp = ops.pack_h2f8(x, st), ops.pack_conv_weight_h3f8(w)
    out = {}
    if up == 2:
        lib.nb_debug_set_up2_v2(1)              # (the 12-row software-pipelined kernel whatever the launch size: on other tile forms in_fmt 3 is in_fmt 1)
    for fmt in (1, 3):
        y = torch.empty([n, co, res, res], device="cuda")
        if up == 1:
            rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), y.data_ptr(), None, None,
                                             0, 0, None, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
        else:
            rc = lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), y.data_ptr(), None, None,
                                             0, 0, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
        if rc != 0:
            lib.nb_debug_set_up2_v2(-1)
        _lib.check(rc, "conv")
        out[fmt] = y.cpu().double()
    lib.nb_debug_set_up2_v2(-1)
    xr = (x * st[:, :, None, None]).half().float()                    # the hi halves as the pack rounds them
    ref_hi = _conv_ref(xr, w.half().float(), torch.ones_like(st), up)
    ref = _conv_ref(x, w, st, up)
    scale = float(ref.abs().max())
    e_hi = float((out[3] - ref_hi).abs().max()) / scale
    e_f16 = float((out[3] - ref).abs().max()) / scale
    e_f8 = float((out[1] - ref).abs().max()) / scale
    print(f"[hi-only up{up} {ci}->{co}@{res}] vs rounded-operand float64 {e_hi:.1e}; vs fp32 operands: f16 {e_f16:.1e}, f8 {e_f8:.1e}")
    assert e_hi <= 5e-6 and e_f8 <= 5e-5 and 5 * e_f8 < e_f16 < 5e-3


@pytest.mark.parametrize("fmt,ci,co,h,w,n", [(1, 128, 64, 128, 128, 32), (1, 384, 128, 64, 64, 32), (0, 64, 64, 48, 32, 40), (1, 32, 32, 26, 64, 70), (1, 64, 96, 24, 32, 9)])
def test_up2v_persistent_workgroups_equal_one_workgroup_per_tile(fmt, ci, co, h, w, n):
    """The 12-row software-pipelined up=2 kernel runs as PERSISTENT workgroups since round 6 (one per CU, each walking its share of the
    launch's tiles, the next tile's first chunk prefetched under the current tile's epilogue).  Same tiles, same per-tile arithmetic:
    fp32 and hand-off outputs must be bit-identical to the one-workgroup-per-tile launch -- at BASELINE's two large shapes (11 and 6
    tiles per workgroup), with ragged last tile rows, with more c_out slices than one, with fewer tiles than CUs (one each) and with a
    tile count that is no multiple of 8 or of the CU count (no XCD renumbering, ragged shares)."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(ci + h + fmt)
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, 2 * h, 2 * w).astype(np.float32)).cuda()
    xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
    wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(wt)
    del x
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    lib.nb_debug_set_up2v_persistent.argtypes, lib.nb_debug_set_up2v_persistent.restype = [ctypes.c_int], None
    lib.nb_debug_set_persistent_wgs_per_cu.argtypes, lib.nb_debug_set_persistent_wgs_per_cu.restype = [ctypes.c_int], None
    res = {}
    try:
        lib.nb_debug_set_up2_v2(1)
        for mode in (0, 1, 2):                            # one workgroup per tile; persistent, 4 workgroups per CU (default); 1 per CU
            lib.nb_debug_set_up2v_persistent(min(mode, 1))
            lib.nb_debug_set_persistent_wgs_per_cu(1 if mode == 2 else 0)
            outs = []
            for rep in range(2):                         # (twice: the second launch starts with the first one's data in the caches)
                y = torch.full([n, co, 2 * h, 2 * w], float("nan"), device="cuda") if fmt == 0 or co <= 64 else None
                out = torch.zeros(ops.h2_shape(n, co, 2 * h, 2 * w), dtype=torch.float16, device="cuda")
                common = (dco.data_ptr(), noise.data_ptr(), 4 * h * w, bias.data_ptr())
                if y is not None:
                    _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, fmt, 0,
                                                           n, h, w, co, 0.2, 1.4142135, 256.0, S), "f32 out")
                _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), co,
                                                       co, fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S), "hand-off out")
                torch.cuda.synchronize()
                outs.append((y, out))
            res[mode] = outs
    finally:
        lib.nb_debug_set_up2v_persistent(-1)
        lib.nb_debug_set_persistent_wgs_per_cu(0)
        lib.nb_debug_set_up2_v2(-1)
    for rep in range(2):
        for mode in (1, 2):
            for a, b in zip(res[0][rep], res[mode][rep]):
                if a is not None:
                    assert torch.equal(a, b)
                    assert bool(torch.isfinite(a.float()).all())


@pytest.mark.parametrize("fmt,ci,co,h,w,n,out", [(1, 128, 128, 128, 128, 32, "handoff"), (1, 64, 64, 256, 256, 32, "torgb"), (1, 64, 64, 64, 64, 40, "f32"),
                                                (0, 48, 64, 32, 64, 70, "handoff"), (1, 32, 192, 32, 32, 33, "handoff"), (1, 64, 64, 32, 64, 9, "torgb"),
                                                # several c_out slices: a workgroup's next item in the SAME slice (32 items per sample: stride 256 keeps the
                                                # slice) and in ANOTHER one (12 per sample) -- the next tile's weights are the same pieces or other ones
                                                (1, 128, 256, 64, 64, 24, "handoff"), (1, 64, 384, 32, 32, 40, "handoff")])
def test_up1_persistent_workgroups_equal_one_workgroup_per_tile(fmt, ci, co, h, w, n, out):
    """The 8-wave up=1 kernel runs as PERSISTENT workgroups since round 6 (one per CU, each walking its share of the launch's tiles, the
    next tile's prologue issued ahead of the current tile's epilogue where that epilogue works from the accumulators).  Same tiles, same
    per-tile arithmetic: hand-off, fused-ToRGB and fp32 outputs must be bit-identical to the one-workgroup-per-tile launch -- at BASELINE's
    two large shapes (8 and 16 tiles per workgroup), with several c_out slices, with item counts that are no multiple of 8 or of the CU
    count, and with fewer items than CUs (one each)."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(ci + h + fmt)
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, h, w).astype(np.float32)).cuda()
    xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
    wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(wt)
    del x
    # fused ToRGB operands (the last layer's form: c_out <= 128, one slice)
    tst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co + 9)).astype(np.float32)).cuda()
    tw = torch.from_numpy((rs.randn(3, co) / np.sqrt(co)).astype(np.float32)).cuda()
    tb, cb = torch.from_numpy(rs.randn(3).astype(np.float32)).cuda(), torch.from_numpy(rs.randn(9).astype(np.float32)).cuda()
    lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
    lib.nb_debug_set_up1_persistent.argtypes, lib.nb_debug_set_up1_persistent.restype = [ctypes.c_int], None
    lib.nb_debug_set_persistent_wgs_per_cu.argtypes, lib.nb_debug_set_persistent_wgs_per_cu.restype = [ctypes.c_int], None
    res = {}
    try:
        for mode in (0, 1, 2):                            # one workgroup per tile; persistent, 4 workgroups per CU (default); 1 per CU
            lib.nb_debug_set_up1_persistent(min(mode, 1))
            lib.nb_debug_set_persistent_wgs_per_cu(1 if mode == 2 else 0)
            outs = []
            for rep in range(2):
                common = (dco.data_ptr(), noise.data_ptr(), h * w, bias.data_ptr())
                if out == "handoff":
                    y = torch.zeros(ops.h2_shape(n, co, h, w), dtype=torch.float16, device="cuda")
                    rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, y.data_ptr(), nst.data_ptr(), co, co, None,
                                                     fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S)
                    got = (y,)
                elif out == "f32":
                    y = torch.full([n, co, h, w], float("nan"), device="cuda")
                    rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, None,
                                                     fmt, 0, n, h, w, co, 0.2, 1.4142135, 256.0, S)
                    got = (y,)
                else:
                    uvs = torch.full([n, 3, h, w], float("nan"), device="cuda")
                    img = torch.full([n, 3, h, w], float("nan"), device="cuda")
                    colors = torch.zeros([n, 3, 3], device="cuda")
                    u8 = torch.zeros([n, h, w, 4], dtype=torch.uint8, device="cuda")
                    t = _lib.NbTorgbArgs()
                    t.styles, t.w, t.bias, t.color_bias = tst.data_ptr(), tw.data_ptr(), tb.data_ptr(), cb.data_ptr()
                    t.logits, t.uvs, t.img, t.colors_out = None, uvs.data_ptr(), img.data_ptr(), colors.data_ptr()
                    t.user_colors, t.sfactor, t.rgba_f32, t.rgba_u8 = None, None, None, u8.data_ptr()
                    t.styles_stride_n, t.render_mode, t.clamp = co + 9, 0, 256.0
                    rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, None, None, 0, 0, ctypes.byref(t),
                                                     fmt, 0, n, h, w, co, 0.2, 1.4142135, 256.0, S)
                    got = (uvs, img, colors, u8)
                _lib.check(rc, "up1")
                torch.cuda.synchronize()
                outs.append(got)
            res[mode] = outs
    finally:
        lib.nb_debug_set_up1_persistent(-1)
        lib.nb_debug_set_persistent_wgs_per_cu(0)
    for rep in range(2):
        for mode in (1, 2):
            for a, b in zip(res[0][rep], res[mode][rep]):
                assert torch.equal(a, b)
                assert bool(torch.isfinite(a.float()).all())
        if out == "torgb":                               # every sample's colors are written (by the sample's lead item), networks.py:462-466
            want = torch.tanh(tst[:, :9] + cb[None, :]).reshape(n, 3, 3)
            assert float((res[1][rep][2] - want).abs().max()) <= 1e-6
