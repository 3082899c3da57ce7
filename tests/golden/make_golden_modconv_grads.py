"""Golden vectors for the modulated-conv gradients (row f4, second slice): the REFERENCE ``modulated_conv2d``
(training/networks.py:30-88) under torch.autograd on CPU (build container only; conv2d_gradfix falls through to
F.conv2d / F.conv_transpose2d there, conv2d_gradfix.py:35-56; upfirdn2d takes its _ref path).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_modconv_grads.py

Cases: up = 1 (flip_weight=True) and up = 2 (flip_weight=False, [1,3,3,1] filter), demodulate on / off, fused and
non-fused forward, with a per-sample noise input; outputs y and dL/dx, dL/dweight, dL/dstyles, dL/dnoise for a random dy.
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
from thirdparty.stylegan2_ada_pytorch.training import networks as ref_networks  # noqa: E402
from torch_utils.ops import upfirdn2d as rup  # noqa: E402


def main():
    out = {}
    rng = np.random.RandomState(4321)
    f = rup.setup_filter([1, 3, 3, 1])
    out["f"] = f.numpy()
    for up, n, ci, co, h in ((1, 2, 12, 10, 8), (2, 2, 10, 12, 4), (1, 2, 40, 36, 8), (2, 1, 36, 40, 4)):
        for demod in (True, False):
            tag = f"up{up}_c{ci}_{'d' if demod else 'n'}"
            x0 = rng.randn(n, ci, h, h).astype(np.float32)
            w0 = (rng.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)
            s0 = (1 + 0.4 * rng.randn(n, ci)).astype(np.float32)
            nz0 = (0.1 * rng.randn(n, 1, h * up, h * up)).astype(np.float32)
            dy0 = rng.randn(n, co, h * up, h * up).astype(np.float32)
            res = {}
            for fused in (True, False):
                x = torch.tensor(x0, requires_grad=True); w = torch.tensor(w0, requires_grad=True)
                s = torch.tensor(s0, requires_grad=True); nz = torch.tensor(nz0, requires_grad=True)
                y = ref_networks.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, up=up, padding=1,
                                                  resample_filter=f if up == 2 else None, demodulate=demod,
                                                  flip_weight=(up == 1), fused_modconv=fused)
                g = torch.autograd.grad(y, [x, w, s, nz], torch.tensor(dy0))
                res[fused] = [y.detach().numpy()] + [t.numpy() for t in g]
            for a, b in zip(res[True], res[False]):           # the reference's two forms agree (SURVEY note A: 7e-7)
                assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), tag
            out.update({tag + "_x": x0, tag + "_w": w0, tag + "_s": s0, tag + "_nz": nz0, tag + "_dy": dy0})
            for name, v in zip(("y", "dx", "dw", "ds", "dnz"), res[True]):
                out[f"{tag}_{name}"] = v
            # path-length style double backward (loss_modified.py:205-221): gradient of |d(y . r)/d styles|^2
            x = torch.tensor(x0, requires_grad=True); w = torch.tensor(w0, requires_grad=True)
            s = torch.tensor(s0, requires_grad=True)
            y = ref_networks.modulated_conv2d(x=x, weight=w, styles=s, noise=torch.tensor(nz0), up=up, padding=1,
                                              resample_filter=f if up == 2 else None, demodulate=demod,
                                              flip_weight=(up == 1), fused_modconv=False)
            gs, = torch.autograd.grad((y * torch.tensor(dy0)).sum(), [s], create_graph=True)
            pl = gs.square().sum()
            g2 = torch.autograd.grad(pl, [x, w, s], allow_unused=True)       # (without demodulation pl does not depend on s)
            out[f"{tag}_pl"] = np.array([float(pl)])
            for name, v, like in zip(("pl_dx", "pl_dw", "pl_ds"), g2, (x0, w0, s0)):
                out[f"{tag}_{name}"] = np.zeros_like(like) if v is None else v.numpy()
    np.savez_compressed(os.path.join(HERE, "modconv_grads.npz"), **out)
    print("modconv_grads.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "modconv_grads.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
