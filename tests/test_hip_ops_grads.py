"""GPU parity (row f4, first slice): the HIP bias_act / upfirdn2d operators with the reference's gradient modes -
forward, dx/db and the second-order terms - against the reference-generated golden vectors
(tests/golden/ops_grads.npz) and against the CPU oracle under torch.autograd on seeded inputs.
fp32 tolerances are written at each check."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ACTS = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def k():
    return load_golden("ops_grads.npz")


def D(a, dev, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(grad)


def close(got, want, tol):
    got = np.zeros_like(want) if got is None else got.detach().cpu().numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float(np.abs(got.astype(np.float64) - np.asarray(want, np.float64)).max())
    assert err <= tol, f"max abs err {err} > {tol}"


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("tag,clamp", [("n", None), ("c", 0.8)])
def test_bias_act_grads_golden(dev, k, act, tag, clamp):
    from brushstroke_engine_amd import ops
    x, b, dy = D(k["ba_x"], dev, True), D(k["ba_b"], dev, True), D(k["ba_dy"], dev, True)
    y = ops.bias_act(x, b, dim=1, act=act, clamp=clamp)
    dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
    d_dy, d_x, d_b = torch.autograd.grad((dx * D(k["ba_ddx"], dev)).sum(), [dy, x, b], allow_unused=True)
    p = f"ba_{act}_{tag}"
    close(y, k[p + "_y"], 2e-6)
    close(dx, k[p + "_dx"], 4e-6)
    close(db, k[p + "_db"], 4e-5)          # sum over 84 elements
    close(d_dy, k[p + "_ddy"], 4e-6)
    close(d_x, k[p + "_d2x"], 4e-6)
    close(d_b, k[p + "_d2b"], 4e-5)


def test_bias_act_last_dim_and_no_grad(dev, k):
    from brushstroke_engine_amd import ops
    x, b = D(k["ba2_x"], dev, True), D(k["ba2_b"], dev, True)
    y = ops.bias_act(x, b, dim=1, act="lrelu", gain=0.7, alpha=0.1)
    dx, db = torch.autograd.grad(y, [x, b], D(k["ba2_dy"], dev))
    close(y, k["ba2_y"], 2e-6); close(dx, k["ba2_dx"], 2e-6); close(db, k["ba2_db"], 2e-5)
    with torch.no_grad():
        assert not ops.bias_act(x, b, act="lrelu").requires_grad
    # gradient only w.r.t. the bias
    y = ops.bias_act(x.detach(), b, dim=1, act="tanh")
    db2, = torch.autograd.grad(y.sum(), [b])
    want = (1 - torch.tanh(x.detach() + b.detach()) ** 2).sum(0)
    close(db2, want.cpu().numpy(), 2e-5)


@pytest.mark.parametrize("act", ACTS)
def test_bias_act_grads_oracle_large(dev, act):
    """Odd sizes (scalar path) and a 16-byte friendly shape (vector path) vs the oracle under autograd."""
    from brushstroke_engine_amd import ops
    from oracle import neube_oracle as orc
    rng = np.random.RandomState(5)
    for shape in ((3, 7, 5, 9), (2, 16, 12, 8)):
        x0 = (rng.randn(*shape) * 2).astype(np.float32); b0 = rng.randn(shape[1]).astype(np.float32)
        dy0 = rng.randn(*shape).astype(np.float32); dd0 = rng.randn(*shape).astype(np.float32)
        res = []
        for mod, to in ((ops, lambda a: D(a, dev, True)), (orc, lambda a: torch.tensor(a, requires_grad=True))):
            x, b, dy = to(x0), to(b0), to(dy0)
            y = mod.bias_act(x, b, dim=1, act=act, gain=1.3, clamp=1.1)
            dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
            dd = to(dd0).detach()
            d_dy, d_x = torch.autograd.grad((dx * dd).sum(), [dy, x], allow_unused=True)
            res.append([y, dx, db, d_dy, d_x])
        for got, want, tol in zip(res[0], res[1], (4e-6, 1e-5, 2e-4, 1e-5, 2e-5)):
            close(got, np.zeros(shape, np.float32) if want is None else want.detach().numpy(), tol)


def _cfg(k, name):
    return ast.literal_eval(str(k[f"up_{name}_cfg"][0]))


@pytest.mark.parametrize("name", list("abcdeg"))
def test_upfirdn2d_grads_golden(dev, k, name):
    from brushstroke_engine_amd import ops
    c = _cfg(k, name)
    x = D(k["up_x"], dev, True)
    y = ops.upfirdn2d(x, D(k["up_" + c["f"]], dev), up=c["up"], down=c["down"], padding=c["padding"],
                      flip_filter=c["flip_filter"], gain=c["gain"])
    dx, = torch.autograd.grad(y, [x], D(k[f"up_{name}_dy"], dev), create_graph=True)
    close(y, k[f"up_{name}_y"], 4e-6)
    close(dx, k[f"up_{name}_dx"], 4e-6)
    # second order: upfirdn2d is linear in x, so d(dx . v)/d(dy) = upfirdn2d(v) with the forward parameters
    v = torch.randn_like(x)
    dy = D(k[f"up_{name}_dy"], dev, True)
    dx2, = torch.autograd.grad(y, [x], dy, create_graph=True)
    g, = torch.autograd.grad((dx2 * v).sum(), [dy])
    want = ops.upfirdn2d(v, D(k["up_" + c["f"]], dev), up=c["up"], down=c["down"], padding=c["padding"],
                         flip_filter=c["flip_filter"], gain=c["gain"])
    close(g, want.detach().cpu().numpy(), 1e-5)


@pytest.mark.parametrize("name,kw", [("filter2d", dict(padding=1)), ("upsample2d", dict(up=2)), ("downsample2d", dict(down=2))])
def test_upfirdn2d_helpers_golden(dev, k, name, kw):
    from brushstroke_engine_amd import ops
    x = D(k["up_x"][:, :, :8, :10], dev, True)
    y = getattr(ops, name)(x, D(k["up_f4"], dev), **kw)
    dx, = torch.autograd.grad(y, [x], D(k[f"uph_{name}_dy"], dev))
    close(y, k[f"uph_{name}_y"], 4e-6)
    close(dx, k[f"uph_{name}_dx"], 4e-6)


def test_setup_filter_separable():
    from brushstroke_engine_amd import ops
    f = ops.setup_filter([1, 2, 4, 7, 7, 4, 2, 1])
    assert f.ndim == 1 and abs(float(f.sum()) - 1) < 1e-6
    f2 = ops.setup_filter([1, 3, 3, 1])
    assert f2.shape == (4, 4)


MG_CASES = [f"up{up}_c{ci}_{d}" for up, ci in ((1, 12), (2, 10), (1, 40), (2, 36)) for d in ("d", "n")]


@pytest.fixture(scope="module")
def mg():
    return load_golden("modconv_grads.npz")


@pytest.mark.parametrize("tag", MG_CASES)
def test_modulated_conv2d_grads_golden(dev, mg, tag):
    """ops.modulated_conv2d under autograd (HIP kernels: fused forward, role-swapped forward / generic strided conv for
    dx, pixel-contraction kernel for the weight / style gradients, upfirdn2d adjoint) against the REFERENCE's gradients."""
    from brushstroke_engine_amd import ops
    up = int(tag[2])
    x, w, s, nz = (D(mg[tag + k_], dev, True) for k_ in ("_x", "_w", "_s", "_nz"))
    y = ops.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=D(mg["f"], dev) if up == 2 else None,
                             demodulate=tag.endswith("_d"), flip_weight=(up == 1))
    g = torch.autograd.grad(y, [x, w, s, nz], D(mg[tag + "_dy"], dev))
    for got, name in zip([y] + list(g), ("y", "dx", "dw", "ds", "dnz")):
        want = mg[f"{tag}_{name}"]
        close(got, want, 5e-5 * max(1.0, float(np.abs(want).max())))


def test_modulated_conv2d_partial_grads_and_no_grad(dev, mg):
    from brushstroke_engine_amd import ops
    tag = "up1_c12_d"
    x, w, s = D(mg[tag + "_x"], dev), D(mg[tag + "_w"], dev, True), D(mg[tag + "_s"], dev)
    y = ops.modulated_conv2d(x, w, s, up=1, padding=1)                 # only the weight needs a gradient, no noise
    dw, = torch.autograd.grad(y, [w], torch.ones_like(y))
    assert dw.shape == w.shape and torch.isfinite(dw).all()
    with torch.no_grad():
        assert not ops.modulated_conv2d(x, w, s, up=1, padding=1).requires_grad


@pytest.mark.parametrize("split_f16", [False, True])
def test_conv2d_and_wgrad_kernels_vs_torch(dev, split_f16):
    """The two gradient building blocks against torch fp64 on odd shapes (ragged channel tiles, stride 2, padding); the
    weight-gradient correlation in its exact-fp32 and its split-f16 form."""
    from brushstroke_engine_amd import ops
    prev = ops.WGRAD_SPLIT_F16
    ops.WGRAD_SPLIT_F16 = split_f16
    try:
        _conv2d_and_wgrad_case(dev, ops)
    finally:
        ops.WGRAD_SPLIT_F16 = prev


def _conv2d_and_wgrad_case(dev, ops):
    rs = np.random.RandomState(3)
    # (the last two: stride 2 without padding at sizes the split-f16 stride-2 kernel takes -- 32-wide and 16-wide output tiles)
    for n, ci, co, h, w_, k, st, pad in ((2, 5, 7, 9, 11, 3, 1, 1), (1, 34, 40, 13, 9, 3, 2, 0), (2, 3, 33, 8, 8, 5, 2, 2), (1, 70, 3, 6, 7, 1, 1, 0),
                                         (8, 24, 40, 33, 65, 3, 2, 0), (16, 20, 70, 33, 33, 3, 2, 0)):
        x = rs.randn(n, ci, h, w_).astype(np.float32); wt = rs.randn(co, ci, k, k).astype(np.float32)
        isc = rs.rand(n, ci).astype(np.float32) + 0.5; osc = rs.rand(n, co).astype(np.float32) + 0.5
        ref = torch.nn.functional.conv2d(torch.tensor(x).double() * torch.tensor(isc).double()[:, :, None, None], torch.tensor(wt).double(),
                                         stride=st, padding=pad) * torch.tensor(osc).double()[:, :, None, None]
        got = ops.conv2d(D(x, dev), D(wt, dev), D(isc, dev), D(osc, dev), stride=st, padding=pad)
        close(got, ref.float().numpy(), 2e-5 * float(ref.abs().max()))
    for n, cu, cv, hv, wv, st, pad in ((2, 5, 7, 6, 9, 1, 1), (1, 40, 33, 5, 4, 2, 0), (3, 33, 2, 8, 8, 1, 1), (2, 64, 130, 16, 40, 1, 1), (8, 128, 128, 4, 4, 1, 1)):
        hu, wu = (hv, wv) if st == 1 else (2 * hv + 1, 2 * wv + 1)
        u = rs.randn(n, cu, hu, wu).astype(np.float32); v = rs.randn(n, cv, hv, wv).astype(np.float32)
        up_ = torch.nn.functional.pad(torch.tensor(u).double(), (pad, pad, pad, pad))
        ref = torch.zeros(n, cu, cv, 3, 3, dtype=torch.float64)
        for a in range(3):
            for b in range(3):
                win = up_[:, :, a:a + st * hv:st, b:b + st * wv:st]
                ref[:, :, :, a, b] = torch.einsum("nuij,nvij->nuv", win, torch.tensor(v).double())
        got = ops.conv2d_wgrad(D(u, dev), D(v, dev), stride=st, padding=pad)
        close(got, ref.float().numpy(), 2e-5 * float(ref.abs().max()))
        # the sum over the samples inside the launch (what a plain convolution's weight gradient takes)
        got = ops._wgrad_launch(D(u, dev), D(v, dev), st, pad, sum_n=True)
        close(got, ref.sum(dim=0).float().numpy(), 2e-5 * float(ref.sum(dim=0).abs().max()))


@pytest.mark.parametrize("tag", MG_CASES)
def test_modulated_conv2d_double_backward_golden(dev, mg, tag):
    """Path-length style second order (loss_modified.py:205-221): gradient of |d(y . r)/d styles|^2 w.r.t. x, weight and
    styles - the create_graph=True mode of ops.modulated_conv2d - against the REFERENCE's autograd."""
    from brushstroke_engine_amd import ops
    up = int(tag[2])
    x, w, s = (D(mg[tag + k_], dev, True) for k_ in ("_x", "_w", "_s"))
    y = ops.modulated_conv2d(x, w, s, noise=D(mg[tag + "_nz"], dev), up=up, padding=1,
                             resample_filter=D(mg["f"], dev) if up == 2 else None, demodulate=tag.endswith("_d"), flip_weight=(up == 1))
    gs, = torch.autograd.grad((y * D(mg[tag + "_dy"], dev)).sum(), [s], create_graph=True)
    pl = gs.square().sum()
    assert abs(float(pl.detach()) - float(mg[tag + "_pl"][0])) <= 2e-4 * float(mg[tag + "_pl"][0])
    g2 = torch.autograd.grad(pl, [x, w, s], allow_unused=True)
    for got, name in zip(g2, ("pl_dx", "pl_dw", "pl_ds")):
        want = mg[f"{tag}_{name}"]
        close(got, want, 5e-4 * max(1e-6, float(np.abs(want).max())))
