"""Golden vectors for the operator gradients (row f4, first slice): run the REFERENCE ``_ref`` ops under
torch.autograd on CPU (build container only) and store inputs + outputs as a small ``.npz`` fixture.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grads.py

bias_act (bias_act.py:93-123): all nine activations x clamp {None, 0.8}: y, first-order (dx, db) for a
random dy, and the second-order terms of g(dy, x, b) = dx . ddx for a random ddx (d_dy, d_x, d_b) - the
quantities ``BiasActCudaGrad.backward`` (bias_act.py:186-204) returns.
upfirdn2d (upfirdn2d.py:168-208): forward + dx for up/down/padding/flip/gain combinations, incl. a separable
1-D filter (the reference's path for >= 8 taps) and the filter2d / upsample2d / downsample2d helpers.
The fixture is data only; the tests that read it never touch /root/reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import thirdparty.stylegan2_ada_pytorch  # noqa: E402,F401
from torch_utils.ops import bias_act as rba  # noqa: E402
from torch_utils.ops import upfirdn2d as rup  # noqa: E402

ACTS = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]


def main():
    out = {}
    rng = np.random.RandomState(1234)
    x0 = (rng.randn(2, 5, 6, 7) * 1.5).astype(np.float32)
    b0 = (rng.randn(5) * 0.5).astype(np.float32)
    dy0 = rng.randn(2, 5, 6, 7).astype(np.float32)
    ddx0 = rng.randn(2, 5, 6, 7).astype(np.float32)
    out.update(ba_x=x0, ba_b=b0, ba_dy=dy0, ba_ddx=ddx0)
    for act in ACTS:
        for tag, clamp in (("n", None), ("c", 0.8)):
            x = torch.tensor(x0, requires_grad=True)
            b = torch.tensor(b0, requires_grad=True)
            dy = torch.tensor(dy0, requires_grad=True)
            y = rba._bias_act_ref(x, b, dim=1, act=act, clamp=clamp)
            dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
            g = (dx * torch.tensor(ddx0)).sum()
            d_dy, d_x, d_b = torch.autograd.grad(g, [dy, x, b], allow_unused=True)
            k = f"ba_{act}_{tag}"
            out[k + "_y"] = y.detach().numpy()
            out[k + "_dx"] = dx.detach().numpy()
            out[k + "_db"] = db.detach().numpy()
            out[k + "_ddy"] = d_dy.numpy()
            out[k + "_d2x"] = np.zeros_like(x0) if d_x is None else d_x.numpy()
            out[k + "_d2b"] = np.zeros_like(b0) if d_b is None else d_b.numpy()
    # a 2-D case with the bias on the last dim (FullyConnectedLayer's use)
    x2 = rng.randn(9, 13).astype(np.float32); b2 = rng.randn(13).astype(np.float32); dy2 = rng.randn(9, 13).astype(np.float32)
    x = torch.tensor(x2, requires_grad=True); b = torch.tensor(b2, requires_grad=True)
    y = rba._bias_act_ref(x, b, dim=1, act="lrelu", gain=0.7, alpha=0.1)
    dx, db = torch.autograd.grad(y, [x, b], torch.tensor(dy2))
    out.update(ba2_x=x2, ba2_b=b2, ba2_dy=dy2, ba2_y=y.detach().numpy(), ba2_dx=dx.numpy(), ba2_db=db.numpy())

    # upfirdn2d
    xu = rng.randn(2, 3, 9, 11).astype(np.float32)
    f2 = rng.rand(4, 3).astype(np.float32)              # asymmetric, non-square 2-D filter [fh=4, fw=3]
    f1 = rup.setup_filter([1, 2, 4, 7, 7, 4, 2, 1]).numpy()       # 8 taps -> separable 1-D
    assert f1.ndim == 1
    f4 = rup.setup_filter([1, 3, 3, 1]).numpy()
    out.update(up_x=xu, up_f2=f2, up_f1=f1, up_f4=f4)
    cases = {
        "a": dict(f="f4", up=2, down=1, padding=[2, 1, 2, 1], flip_filter=False, gain=4.0),
        "b": dict(f="f2", up=1, down=2, padding=[1, 2, 0, 3], flip_filter=False, gain=1.0),
        "c": dict(f="f2", up=[2, 3], down=[3, 2], padding=[3, 1, 2, 4], flip_filter=True, gain=2.5),
        "d": dict(f="f1", up=2, down=1, padding=[4, 3, 4, 3], flip_filter=False, gain=4.0),
        "e": dict(f="f1", up=1, down=2, padding=[3, 3, 3, 3], flip_filter=True, gain=1.0),
        "g": dict(f="f2", up=1, down=1, padding=[-1, 2, 1, -2], flip_filter=False, gain=1.0),
    }
    fs = {"f2": f2, "f1": f1, "f4": f4}
    for name, c in cases.items():
        x = torch.tensor(xu, requires_grad=True)
        y = rup._upfirdn2d_ref(x, torch.tensor(fs[c["f"]]), up=c["up"], down=c["down"], padding=c["padding"],
                               flip_filter=c["flip_filter"], gain=c["gain"])
        dy = torch.tensor(rng.randn(*y.shape).astype(np.float32))
        dx, = torch.autograd.grad(y, [x], dy)
        out[f"up_{name}_y"] = y.detach().numpy(); out[f"up_{name}_dy"] = dy.numpy(); out[f"up_{name}_dx"] = dx.numpy()
        out[f"up_{name}_cfg"] = np.array([str(c)])
    for name, fn, kw in (("filter2d", rup.filter2d, dict(padding=1)), ("upsample2d", rup.upsample2d, dict(up=2)),
                         ("downsample2d", rup.downsample2d, dict(down=2))):
        x = torch.tensor(xu[:, :, :8, :10].copy(), requires_grad=True)
        y = fn(x, torch.tensor(f4), impl="ref", **kw)
        dy = torch.tensor(rng.randn(*y.shape).astype(np.float32))
        dx, = torch.autograd.grad(y, [x], dy)
        out[f"uph_{name}_y"] = y.detach().numpy(); out[f"uph_{name}_dy"] = dy.numpy(); out[f"uph_{name}_dx"] = dx.numpy()
    np.savez_compressed(os.path.join(HERE, "ops_grads.npz"), **out)
    print("ops_grads.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "ops_grads.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
