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


@pytest.mark.parametrize("shape", [(2, 144, 128, 32, 32), (1, 384, 128, 64, 64), (1, 128, 64, 128, 128), (2, 40, 48, 16, 32)])
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
