"""Golden vectors for the discriminator (row f4): the REFERENCE ``training.networks.Discriminator`` (resnet, c_dim=0) on
CPU (build container only) with randomised parameters: logits, gradients of sum(logits) w.r.t. the image and every
parameter, and the R1 term's double backward (gradient of sum((d logits / d img)^2) w.r.t. every parameter,
``loss_modified.py`` Dreg phase).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_discriminator.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import thirdparty.stylegan2_ada_pytorch  # noqa: E402,F401
from thirdparty.stylegan2_ada_pytorch.training import networks as ref_networks  # noqa: E402


def main():
    torch.manual_seed(7)
    kw = dict(c_dim=0, img_resolution=32, img_channels=4, channel_base=512, channel_max=24, conv_clamp=256)
    with contextlib.redirect_stdout(io.StringIO()):
        D = ref_networks.Discriminator(**kw)
    rng = np.random.RandomState(99)
    with torch.no_grad():
        for name, p in D.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.from_numpy((0.2 * rng.randn(*p.shape)).astype(np.float32)))
    out = {"kw": np.array([str(kw)])}
    for k, v in D.state_dict().items():
        out["sd." + k] = v.numpy()
    img0 = rng.randn(8, 4, 32, 32).astype(np.float32)
    img = torch.tensor(img0, requires_grad=True)
    logits = D(img, None)
    params = [p for _, p in D.named_parameters()]
    names = [n for n, _ in D.named_parameters()]
    g = torch.autograd.grad(logits.sum(), [img] + params, create_graph=True)
    r1 = g[0].square().sum()
    g2 = torch.autograd.grad(r1, params, allow_unused=True)
    out.update(img=img0, logits=logits.detach().numpy(), dimg=g[0].detach().numpy(), r1=np.array([float(r1)]))
    for n, a, b in zip(names, g[1:], g2):
        out["g." + n] = a.detach().numpy()
        out["r1." + n] = np.zeros_like(a.detach().numpy()) if b is None else b.numpy()
    np.savez_compressed(os.path.join(HERE, "discriminator_r32.npz"), **out)
    print("discriminator_r32.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "discriminator_r32.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
