"""GPU parity: each HIP operator (through the C ABI) against the reference-generated KATs
(tests/golden/ops_kat.npz) and against the CPU oracle on seeded inputs.  fp32 tolerances are
written at each check."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def D(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, want, tol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= tol, f"max abs err {err} > {tol}"


def test_bias_act_kats(dev):
    from brushstroke_engine_amd import ops
    k = load_golden("ops_kat.npz")
    x, b = D(k["ba_x"], dev), D(k["ba_b"], dev)
    for act in ("lrelu", "linear", "tanh"):
        close(ops.bias_act(x, b, act=act), k[f"ba_{act}_n"], 1e-6)
        close(ops.bias_act(x, b, act=act, clamp=1.5), k[f"ba_{act}_c"], 1e-6)
    close(ops.bias_act(x, b, act="lrelu", gain=np.sqrt(2) * 0.5, clamp=128.0), k["ba_lrelu_gain"], 1e-6)
    close(ops.bias_act(D(k["ba2_x"], dev), D(k["ba2_b"], dev), dim=1, act="tanh"), k["ba2_tanh"], 1e-6)
    # vectorised path (size % 4 == 0, bias stride % 4 == 0) and no-bias path, against the oracle
    from oracle import neube_oracle as orc
    rs = np.random.RandomState(0)
    xx = rs.randn(2, 8, 16, 16).astype(np.float32) * 4
    bb = rs.randn(8).astype(np.float32)
    close(ops.bias_act(D(xx, dev), D(bb, dev), act="lrelu", clamp=3.0),
          orc.bias_act(torch.from_numpy(xx), torch.from_numpy(bb), act="lrelu", clamp=3.0), 1e-6)
    close(ops.bias_act(D(xx, dev), None, act="sigmoid"), orc.bias_act(torch.from_numpy(xx), None, act="sigmoid"), 1e-6)
    with pytest.raises(AssertionError):
        ops.bias_act(D(xx, dev), D(bb[:3], dev))
    with pytest.raises(RuntimeError):
        ops.bias_act(torch.from_numpy(xx), None)       # CPU tensor: no CPU path


def test_upfirdn2d_configurations_vs_oracle(dev):
    """Every specialised launch of nb_upfirdn2d_f32 (4x4 / 1x1 / run-time filters at up / down 1 or 2 per axis, the separable
    1-D passes) and the generic kernel behind them, on odd sizes with positive, zero and negative padding, against the oracle."""
    from brushstroke_engine_amd import ops
    from oracle import neube_oracle as orc
    rs = np.random.RandomState(5)
    f44 = ops.setup_filter((1, 3, 3, 1), device=dev)
    f12 = torch.from_numpy(rs.randn(12).astype(np.float32)).to(dev)
    f35 = torch.from_numpy(rs.randn(3, 5).astype(np.float32)).to(dev)
    cases = [(f44, 1, 1, [2, 2, 2, 2], False, 1.0), (f44, 1, 1, [1, 1, 1, 1], True, 4.0), (f44, 2, 1, [2, 1, 2, 1], False, 4.0),
             (f44, 1, 2, [1, 1, 1, 1], False, 1.0), (f44, 1, 2, [2, 2, 2, 2], True, 1.0), (None, 2, 1, [0, -1, 0, -1], False, 1.0),
             (f12, 2, 1, [6, 5, 6, 5], False, 4.0), (f12, 1, 2, [3, 3, 3, 3], True, 1.0), (f35, 1, 1, [2, 1, 0, 3], False, 1.5),
             (f35, 2, 1, [-1, 2, 1, 0], True, 1.0), (f35, 1, 2, [1, 1, 1, 1], False, 1.0), (f35, 3, 2, [2, 2, 2, 2], False, 1.0),
             (f44, (2, 1), 1, [2, 1, 1, 1], False, 2.0), (f44, 1, (1, 2), [1, 1, 1, 1], False, 1.0),
             # factors >= 4 (they would alias the 2-bit fields of the specialised kernels' dispatch key: down = (1, 5) read as <1,1,2,1>)
             (f44, 1, (1, 5), [2, 2, 2, 2], False, 1.0), (f44, 1, (5, 1), [2, 2, 2, 2], False, 1.0), (f44, 4, 1, [3, 3, 3, 3], False, 16.0),
             (f35, (2, 1), (5, 1), [2, 2, 2, 2], True, 1.0)]
    for shape in ((2, 3, 9, 13), (1, 5, 33, 17)):
        x = rs.randn(*shape).astype(np.float32)
        for f, up, down, pad, flip, gain in cases:
            want = orc.upfirdn2d(torch.from_numpy(x), torch.ones(1, 1) if f is None else f.cpu(), up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
            got = ops.upfirdn2d(D(x, dev), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
            assert tuple(got.shape) == tuple(want.shape), (shape, up, down, pad)
            close(got, want, 2e-5 * max(1.0, float(want.abs().max())))


def test_upfirdn2d_kats(dev):
    from brushstroke_engine_amd import ops
    k = load_golden("ops_kat.npz")
    f = ops.setup_filter((1, 3, 3, 1), device=dev)
    close(f, k["fir_f"], 0)
    x = D(k["fir_x"], dev)
    close(ops.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4), k["fir_pad1_gain4"], 1e-6)
    close(ops.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4), k["fir_up2"], 1e-6)
    close(ops.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1]), k["fir_down2"], 1e-6)
    close(ops.upfirdn2d(x, D(k["fir_f_ragged"], dev), padding=[1, 0, 2, -1], flip_filter=True, gain=1.5),
          k["fir_ragged_flip"], 1e-6)


def test_modulated_conv2d_kats(dev):
    """Reference KATs (6 -> 5 channels at 7x7) embedded into a shape the MFMA kernel accepts: the
    7x7 image is zero-padded to 8x8 (so the reference's zero padding at rows/cols 7 is reproduced by
    real zeros) and the weight to 32 output channels."""
    from brushstroke_engine_amd import ops
    k = load_golden("ops_kat.npz")
    xin = np.zeros((2, 6, 8, 8), np.float32)
    xin[:, :, :7, :7] = k["mc_x"]
    x, s = D(xin, dev), D(k["mc_s"], dev)
    w = np.zeros((32, 6, 3, 3), np.float32)
    w[:5] = k["mc_w"]
    w = D(w, dev)
    f = ops.setup_filter((1, 3, 3, 1), device=dev)
    for up in (1, 2):
        for demod in (True, False):
            y = ops.modulated_conv2d(x, w, s, noise=None, up=up, padding=1, resample_filter=f, demodulate=demod,
                                     flip_weight=(up == 1))
            close(y[:, :5, :7 * up, :7 * up], k[f"mc_up{up}_d{int(demod)}_n0_f1"], 2e-5)
            if not demod:
                assert float(y[:, 5:].abs().max()) == 0.0


@pytest.mark.parametrize("up", [1, 2])
@pytest.mark.parametrize("shape", [(2, 36, 32, 4), (3, 40, 32, 8), (2, 32, 16, 16), (1, 128, 128, 32), (2, 16, 64, 64),
                                   (1, 144, 128, 32), (1, 64, 64, 128)])
def test_modconv_vs_oracle(dev, up, shape):
    """Fused layer (conv + demod + noise + bias + lrelu + clamp) against the CPU oracle; ragged channel
    counts (36, 40), every tile shape from 4x4 to 128x128, with a split geometry input."""
    from brushstroke_engine_amd import ops
    from oracle import neube_oracle as orc
    n, ic, oc, h = shape
    if up == 2 and h > 64:
        h = 64
    rs = np.random.RandomState(ic * 7 + h + up)
    c2 = 4 if ic % 16 else (16 if ic > 16 else 0)
    c1 = ic - c2
    x = rs.randn(n, ic, h, h).astype(np.float32)
    w = rs.randn(oc, ic, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32)
    b = (0.1 * rs.randn(oc)).astype(np.float32)
    noise = (0.1 * rs.randn(n, 1, h * up, h * up)).astype(np.float32)
    f = orc.setup_filter((1, 3, 3, 1))
    T = torch.from_numpy
    want = orc.modulated_conv2d(T(x), T(w), T(s), noise=T(noise), up=up, padding=1, resample_filter=f,
                                flip_weight=(up == 1))
    want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=2.5)
    got = ops.modulated_conv2d(D(x[:, :c1], dev), D(w, dev), D(s, dev), noise=D(noise, dev), up=up, padding=1,
                               resample_filter=ops.setup_filter((1, 3, 3, 1), device=dev), flip_weight=(up == 1),
                               x2=D(x[:, c1:], dev) if c2 else None, bias=D(b, dev), act_gain=np.sqrt(2), act_clamp=2.5,
                               fuse_bias_act=True)
    close(got, want, 5e-5)


def test_blend_vs_oracle(dev):
    from brushstroke_engine_amd import ops
    rs = np.random.RandomState(1)
    x = rs.randn(3, 8, 16, 16).astype(np.float32)
    for nf, na in ((1, 1), (3, 1), (3, 3)):
        f = rs.randn(nf, 8, 16, 16).astype(np.float32)
        a = rs.rand(na, 1, 16, 16).astype(np.float32)
        close(ops.blend(D(f, dev), D(a, dev), D(x, dev)), a * f + (1 - a) * x, 1e-6)
