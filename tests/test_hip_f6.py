"""GPU parity of the "f6" operand format (round 5: the correction products of the split-f16 scheme on block-scaled fp6 (e2m3) MFMAs,
one E8M0 scale per pixel and 16-channel chunk): the packers against a host restatement of the format, the kernels against float64 and
against the f8 kernels.  north_star tolerance: 1e-3 max abs on pixels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv_ref(x, w, st, up):
    xm = (x * st[:, :, None, None]).double().cpu()
    if up == 1:
        return torch.nn.functional.conv2d(xm, w.double().cpu(), padding=1)
    from oracle import neube_oracle as orc
    f = orc.setup_filter([1, 3, 3, 1], dtype=torch.float64)
    return orc.conv2d_resample(xm, w.double().cpu(), f=f, up=2, padding=1, flip_weight=False)


def test_f6_pack_matches_host_format():
    """nb_pack_h2f6_f32 (v_cvt_scalef32_2xpk16_fp6_f32 on the device) == the format as ops.e2m3_encode / f6_block_exponent restate it on
    the host: hi slots bit for bit, scale byte = exponent of the chunk's largest |x| minus 2, every field = RNE(value / scale)."""
    from brushstroke_engine_amd import ops
    rs = np.random.RandomState(3)
    n, c, h, w = 2, 48, 16, 32
    x = torch.from_numpy((rs.randn(n, c, h, w) * np.exp(rs.randn(n, c, 1, 1))).astype(np.float32)).cuda()
    x[0, :16, 0, 0] = 0.0                                     # an all-zero chunk
    x[0, 16:32, 0, 1] = torch.tensor([7.6, -7.9, 4.0, 3.99] * 4).cuda()      # top of the range: (7.5, 8) saturates
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, c)).astype(np.float32)).cuda()
    t = ops.pack_h2f6(x, st)
    hi, xl, xv, sc = ops.unpack_h2f6(t, c)
    v = (x * st[:, :, None, None]).float()
    hi_want = v.half().float()
    assert torch.equal(hi, hi_want)
    m = v.abs().reshape(n, c // 16, 16, h, w).amax(dim=2)
    e = ops.f6_block_exponent(m)
    nz = m > 0
    assert torch.equal(torch.log2(sc)[nz], e[nz])
    S = torch.exp2(e).repeat_interleave(16, dim=1)
    want_x = ops.e2m3_decode(ops.e2m3_encode(v / S)) * S
    want_xl = ops.e2m3_decode(ops.e2m3_encode((v - hi_want) * 2048.0 / S)) * S
    assert torch.equal(xv, want_x) and torch.equal(xl, want_xl)
    # the value the fields stand for: x to ~3 mantissa bits relative to the chunk's maximum
    assert float((xv - v).abs().max() / v.abs().max()) < 0.07


@pytest.mark.parametrize("up,ci,co,res", [(1, 64, 64, 64), (1, 128, 128, 32), (1, 144, 96, 64), (1, 32, 64, 32),
                                           (2, 128, 64, 128), (2, 384, 128, 64), (2, 144, 128, 64), (2, 64, 32, 256)])
def test_f6_kernels_vs_float64(up, ci, co, res):
    """One layer, f6 operands, fp32 output: error against float64 at the level of the correction terms (2^-11 x 2^-4 relative per
    product, relative to the chunk's largest element): within 2x of the f8 kernel's and ~15x below a plain f16 evaluation."""
    from brushstroke_engine_amd import _lib, ops
    rs = np.random.RandomState(up * 1000 + ci + co)
    n = 2
    hin = res if up == 1 else res // 2
    lib = _lib.lib()
    x = torch.from_numpy((rs.randn(n, ci, hin, hin) * 2).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    dco, bias = torch.ones(n, co, device="cuda"), torch.zeros(co, device="cuda")
    ref = _conv_ref(x, w, st, up)
    S = torch.cuda.current_stream().cuda_stream
    errs = {}
    if up == 2:
        lib.nb_debug_set_up2_v2(1)              # the 12-row software-pipelined kernel (the up = 2 form that takes f6), whatever the launch size
    try:
        for fmt, pack_x, pack_w in ((1, ops.pack_h2f8, ops.pack_conv_weight_h3f8), (2, ops.pack_h2f6, ops.pack_conv_weight_h3f6)):
            xh, wp = pack_x(x, st), pack_w(w)
            y = torch.empty([n, co, res, res], device="cuda")
            if up == 1:
                rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(),
                                                 y.data_ptr(), None, None, 0, 0, None, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
            else:
                rc = lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(),
                                                 y.data_ptr(), None, None, 0, 0, fmt, 0, n, hin, hin, co, 1.0, 1.0, -1.0, S)
            _lib.check(rc, "conv")
            errs[fmt] = float((y.cpu().double() - ref).abs().max())
    finally:
        if up == 2:
            lib.nb_debug_set_up2_v2(-1)
    scale = float(ref.abs().max())
    print(f"[f6 kernel up{up} {ci}->{co}@{res}] max err / max |ref|: f8 {errs[1] / scale:.2e}  f6 {errs[2] / scale:.2e}")
    assert errs[1] <= 4e-5 * scale and errs[2] <= 8e-5 * scale, (errs, scale)
