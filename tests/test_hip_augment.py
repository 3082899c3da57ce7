"""GPU parity of the augmentation pipeline (row f4) against the REFERENCE AugmentPipe in its deterministic
``debug_percentile`` mode (tests/golden/augment.npz), plus: identity at p = 0, twice-differentiable w.r.t. the images
(what the R1 penalty needs behind the augmentation)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

BGC = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
CONFIGS = {"bgc": BGC, "bgcfc": dict(BGC, imgfilter=1, cutout=1), "color": dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
           "geom": dict(scale=1, rotate=1, aniso=1, xfrac=1), "filter": dict(imgfilter=1, imgfilter_bands=[1, 0, 1, 1])}


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("pct", [15, 60, 85])
def test_augment_matches_reference(name, pct):
    from brushstroke_engine_amd.augment import AugmentPipe
    g = load_golden("augment.npz")
    pipe = AugmentPipe(**CONFIGS[name]).to("cuda")
    for key in ("img3", "img1"):
        want = g[f"{name}_{key}_{pct}"]
        got = pipe(torch.from_numpy(g[key]).cuda(), debug_percentile=pct / 100).cpu().numpy()
        assert got.shape == want.shape
        err = float(np.abs(got - want).max())
        assert err <= 2e-4 * max(1.0, float(np.abs(want).max())), (name, key, pct, err)


def test_augment_identity_at_p0_and_double_backward():
    from brushstroke_engine_amd.augment import AugmentPipe
    torch.manual_seed(0)
    x = torch.randn(3, 3, 32, 32, device="cuda")
    pipe = AugmentPipe(**BGC).to("cuda")
    pipe.p.fill_(0.0)
    y = pipe(x)                                           # nothing selected: only the resampling round trip remains
    assert float((y - x).abs().max()) <= 0.15 and float((y - x).abs().mean()) <= 0.02
    pipe.p.fill_(0.7)
    w = torch.randn(3, device="cuda", requires_grad=True)
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(1)
    out = pipe(xr * w[None, :, None, None])
    grad, = torch.autograd.grad(out.square().sum(), [xr], create_graph=True)
    g2, = torch.autograd.grad(grad.square().sum(), [w])
    assert torch.isfinite(g2).all() and float(g2.abs().max()) > 0


def test_gan_loss_with_augment_pipe_and_ada_update():
    """GanLoss routes the discriminator's inputs through the augmentation pipe (run_D of the reference): all phases, incl.
    R1 through the augmented real images (double backward through the resampling), stay finite; ada_update moves p by the
    sign rule of the training loop."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.augment import AugmentPipe
    from brushstroke_engine_amd.training import TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict
    dev = torch.device("cuda:0")
    cfg = cfgmod.tiny_config(32)
    G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 5), dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(32, 3, channel_base=512, channel_max=24, seed=3), 32, 3, channel_base=512,
                               channel_max=24, conv_clamp=256, device=dev)
    pipe = AugmentPipe(**BGC).to(dev)
    pipe.p.fill_(0.5)
    loss = GanLoss(G, D, augment_pipe=pipe)
    z = torch.from_numpy(synthetic.batch_z(cfg, 4, 3)).float().to(dev)
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, 4, 7)]
    real = torch.tanh(torch.randn(4, 3, 32, 32, device=dev))
    for phase in ("Gmain", "Dmain", "Dreg"):
        st = loss.accumulate_gradients(phase, real, geom, z)
        assert all(np.isfinite(v) for v in st.values()), (phase, st)
    assert all(torch.isfinite(p.grad).all() for p in D.parameters())
    p0 = float(pipe.p)
    p1 = loss.ada_update(ada_target=0.6, batch_size=4, ada_interval=4, ada_kimg=0.1)
    assert p1 != p0 and p1 >= 0
