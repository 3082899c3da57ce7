"""Row f4, first slice (CPU): the oracle's bias_act / upfirdn2d - forward and, through torch.autograd, first- and
second-order gradients - against golden vectors produced by the reference's ``_ref`` ops
(tests/golden/make_golden_grads.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import neube_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ops_grads.npz")
ACTS = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]


@pytest.fixture(scope="module")
def k():
    return np.load(GOLD)


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("tag,clamp", [("n", None), ("c", 0.8)])
def test_oracle_bias_act_grads(k, act, tag, clamp):
    x = torch.tensor(k["ba_x"], requires_grad=True); b = torch.tensor(k["ba_b"], requires_grad=True)
    dy = torch.tensor(k["ba_dy"], requires_grad=True)
    y = orc.bias_act(x, b, dim=1, act=act, clamp=clamp)
    dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
    d_dy, d_x, d_b = torch.autograd.grad((dx * torch.tensor(k["ba_ddx"])).sum(), [dy, x, b], allow_unused=True)
    p = f"ba_{act}_{tag}"
    z = lambda t, like: np.zeros_like(like) if t is None else t.detach().numpy()
    for got, name in ((y, "y"), (dx, "dx"), (db, "db"), (d_dy, "ddy")):
        np.testing.assert_allclose(got.detach().numpy(), k[f"{p}_{name}"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(z(d_x, k["ba_x"]), k[f"{p}_d2x"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(z(d_b, k["ba_b"]), k[f"{p}_d2b"], rtol=0, atol=2e-5)


def _cfg(k, name):
    return ast.literal_eval(str(k[f"up_{name}_cfg"][0]))


@pytest.mark.parametrize("name", list("abcdeg"))
def test_oracle_upfirdn2d_grads(k, name):
    c = _cfg(k, name)
    x = torch.tensor(k["up_x"], requires_grad=True)
    y = orc.upfirdn2d(x, torch.tensor(k["up_" + c["f"]]), up=c["up"], down=c["down"], padding=c["padding"],
                      flip_filter=c["flip_filter"], gain=c["gain"])
    dx, = torch.autograd.grad(y, [x], torch.tensor(k[f"up_{name}_dy"]))
    np.testing.assert_allclose(y.detach().numpy(), k[f"up_{name}_y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dx.numpy(), k[f"up_{name}_dx"], rtol=0, atol=2e-6)


MG = os.path.join(os.path.dirname(__file__), "golden", "modconv_grads.npz")
MG_CASES = [f"up{up}_c{ci}_{d}" for up, ci in ((1, 12), (2, 10), (1, 40), (2, 36)) for d in ("d", "n")]


@pytest.mark.parametrize("tag", MG_CASES)
@pytest.mark.parametrize("fused", [True, False])
def test_oracle_modulated_conv2d_grads(tag, fused):
    """The oracle's modulated_conv2d under autograd (both forms) against the reference's gradients."""
    k = np.load(MG)
    up = int(tag[2])
    x = torch.tensor(k[tag + "_x"], requires_grad=True); w = torch.tensor(k[tag + "_w"], requires_grad=True)
    s = torch.tensor(k[tag + "_s"], requires_grad=True); nz = torch.tensor(k[tag + "_nz"], requires_grad=True)
    y = orc.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=torch.tensor(k["f"]) if up == 2 else None,
                             demodulate=tag.endswith("_d"), flip_weight=(up == 1), fused_modconv=fused)
    g = torch.autograd.grad(y, [x, w, s, nz], torch.tensor(k[tag + "_dy"]))
    for got, name in zip([y] + list(g), ("y", "dx", "dw", "ds", "dnz")):
        want = k[f"{tag}_{name}"]
        np.testing.assert_allclose(got.detach().numpy(), want, rtol=0, atol=3e-5 * max(1.0, float(np.abs(want).max())))
