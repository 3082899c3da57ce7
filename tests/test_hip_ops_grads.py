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
                                         (8, 24, 40, 33, 65, 3, 2, 0), (16, 20, 70, 33, 33, 3, 2, 0),
                                         # odd-sized stride-1 correlations: embedded in a tile-able canvas for the split-f16 kernels
                                         (2, 20, 24, 63, 67, 3, 1, 2), (4, 16, 8, 49, 50, 3, 1, 0), (2, 8, 9, 70, 70, 3, 1, 1)):
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


def test_absmax_slots_and_ranged_pack(dev):
    """The training path's range scaling: nb_absmax_f32 leaves max|.| of up to three tensors in two device words (16-byte
    aligned and unaligned inputs, ragged counts, an absent operand), and nb_pack_h2_ranged_f32 = nb_pack_h2_f32 with the power
    of two derived from them (2^floor(log2(target / (max|x| max|s|))): checked against torch), plus the rescaled coefficients."""
    from brushstroke_engine_amd import ops, _lib
    rs = np.random.RandomState(11)
    a = torch.from_numpy(rs.randn(3, 5, 7, 11).astype(np.float32) * 37).to(dev)
    big = torch.from_numpy(rs.randn(1 << 20).astype(np.float32)).to(dev)
    b = big[1:1 + 100003]                                   # 4-byte aligned only
    c = torch.from_numpy(rs.randn(2, 9).astype(np.float32) * 1e-3).to(dev)
    got = ops._absmax_slots(a, b, c).view(torch.float32).cpu().numpy()
    assert got[0] == max(float(a.abs().max()), float(b.abs().max())) and got[1] == float(c.abs().max())
    got = ops._absmax_slots(big).view(torch.float32).cpu().numpy()
    assert got[0] == float(big.abs().max()) and got[1] == 0.0
    # ranged pack vs the plain pack with the scale spelled out
    n, c1, c2, h, w = 2, 12, 5, 16, 32
    x = torch.from_numpy(rs.randn(n, c1, h, w).astype(np.float32) * 300).to(dev)
    x2 = torch.from_numpy(rs.randn(n, c2, h, w).astype(np.float32) * 1e-2).to(dev)
    st = torch.from_numpy(rs.uniform(0.2, 3.0, (n, c1 + c2)).astype(np.float32)).to(dev)
    dco = torch.from_numpy(rs.uniform(0.5, 2.0, (n, 24)).astype(np.float32)).to(dev)
    slots = ops._absmax_slots(x, x2, st)
    mx = max(float(x.abs().max()), float(x2.abs().max())) * float(st.abs().max())
    k = 2.0 ** np.floor(np.log2(16384.0 / mx))
    out = torch.empty(ops.h2_shape(n, c1 + c2, h, w), dtype=torch.float16, device=dev)
    dco_out = torch.empty_like(dco)
    _lib.check(_lib.lib().nb_pack_h2_ranged_f32(x.data_ptr(), c1, x2.data_ptr(), c2, st.data_ptr(), out.data_ptr(), n, h * w, slots.data_ptr(),
                                               16384.0, dco.data_ptr(), dco_out.data_ptr(), dco.numel(), torch.cuda.current_stream().cuda_stream), "ranged")
    want = ops.pack_h2(x, st * float(k), x2)
    assert torch.equal(out, want)
    assert torch.equal(dco_out, dco / float(k))
    assert float(ops.unpack_h2(out, c1 + c2).abs().max()) <= 16384.0 * 1.001


def test_conv2d_down2_is_fir_plus_strided_conv_with_the_same_gradients(dev):
    """``ops.conv2d_down2`` (one operator: its input gradient is one launch of the fused up=2 kernel) against the composition it
    replaces -- ``conv2d(upfirdn2d(x, f, padding=2), w, stride=2)`` on the generic differentiable operators: forward, dx, dw, and
    the R1-type double backward (gradient of |dy/dx|^2 w.r.t. the weight), at a size the split-f16 kernels take and at a small one."""
    from brushstroke_engine_amd import ops
    rs = np.random.RandomState(12)
    f = ops.setup_filter((1, 3, 3, 1), device=dev)
    for n, ci, co, h in ((8, 32, 48, 64), (2, 8, 24, 16)):
        x0 = rs.randn(n, ci, h, h).astype(np.float32)
        w0 = (rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)
        g0 = rs.randn(n, co, h // 2, h // 2).astype(np.float32)
        res = []
        for fused in (True, False):
            x, w = D(x0, dev, True), D(w0, dev, True)
            y = ops.conv2d_down2(x, w, f) if fused else ops.conv2d(ops.upfirdn2d(x, f, padding=[2, 2, 2, 2]), w, stride=2, padding=0)
            dx, dw = torch.autograd.grad(y, [x, w], D(g0, dev), create_graph=False)
            # second order: d/dw of sum((dy.sum()/dx)^2) as the R1 penalty forms it
            x2, w2 = D(x0, dev, True), D(w0, dev, True)
            y2 = ops.conv2d_down2(x2, w2, f) if fused else ops.conv2d(ops.upfirdn2d(x2, f, padding=[2, 2, 2, 2]), w2, stride=2, padding=0)
            gx, = torch.autograd.grad(y2.sum(), [x2], create_graph=True)
            ddw, = torch.autograd.grad(gx.square().sum(), [w2])
            res.append((y.detach(), dx, dw, ddw))
        for got, want, name in zip(res[0], res[1], ("y", "dx", "dw", "ddw")):
            tol = 3e-5 * max(1e-6, float(want.abs().max()))
            close(got, want.cpu().numpy(), tol)
