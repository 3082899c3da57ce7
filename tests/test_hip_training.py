"""GPU parity of the differentiable generator (row f4): forward against the fused inference generator, gradients w.r.t.
parameters, latents and geometry features against the CPU oracle under torch.autograd on the same seeded inputs."""
import numpy as np
import pytest
import torch

from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic

pytestmark = pytest.mark.gpu


def _setup(res=32, n=2, seed=5):
    cfg = cfgmod.tiny_config(res)
    sd = wmod.random_state_dict(cfg, seed)
    z = synthetic.batch_z(cfg, n, 3).astype(np.float32)
    geom = [g.astype(np.float32) for g in synthetic.geom_features(cfg, n, 7)]
    pos = synthetic.positions(cfg, n, 9)
    return cfg, sd, z, geom, pos


@pytest.mark.parametrize("with_pos", [False, True])
def test_trainable_forward_equals_inference_generator(with_pos):
    from brushstroke_engine_amd.networks import Generator
    from brushstroke_engine_amd.training import TrainableGenerator
    cfg, sd, z, geom, pos = _setup()
    dev = torch.device("cuda:0")
    T = TrainableGenerator(cfg, sd, dev)
    G = Generator(cfg, sd, conv_mode="f32").to(dev)
    zt, gt = torch.from_numpy(z).to(dev), [torch.from_numpy(g).to(dev) for g in geom]
    pt = torch.from_numpy(pos).to(dev) if with_pos else None
    with torch.no_grad():
        img, dbg = T(zt, None, gt, positions=pt, return_debug_data=True, noise_mode="const")
    ref, rdbg = G(zt, None, gt, positions=pt, return_debug_data=True, noise_mode="const")
    assert float((img - ref).abs().max()) <= 2e-5 and float((dbg["uvs"] - rdbg["uvs"]).abs().max()) <= 2e-5


def test_trainable_gradients_match_oracle():
    """A scalar loss on the image: gradients of every parameter, of z and of the geometry features, HIP autograd path
    vs the oracle (plain torch CPU ops) - tolerance 2e-4 relative to each gradient's own scale."""
    from brushstroke_engine_amd.training import TrainableGenerator
    from oracle import neube_oracle as orc
    cfg, sd, z, geom, pos = _setup()
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(1)
    target = rs.randn(z.shape[0], 3, cfg.img_resolution, cfg.img_resolution).astype(np.float32)
    npos = ((pos % cfg.img_resolution) / (cfg.img_resolution - 1)).astype(np.float32)
    # oracle
    O = orc.OracleGenerator(cfg, sd)
    train_keys = [k for k in O.sd if k.endswith((".weight", ".bias", ".noise_strength", ".const", ".color_bias"))]
    for k in train_keys:
        O.sd[k].requires_grad_(True)
    zo = torch.tensor(z, requires_grad=True)
    go = [torch.tensor(g, requires_grad=True) for g in geom]
    ws = O.mapping(zo)
    img_o, _ = O.synthesis(ws, go, return_debug_data=True, norm_noise_positions=npos)
    loss_o = ((img_o - torch.tensor(target)) ** 2).mean()
    grads_o = torch.autograd.grad(loss_o, [O.sd[k] for k in train_keys] + [zo] + go, allow_unused=True)
    # HIP
    T = TrainableGenerator(cfg, sd, dev)
    zt = torch.tensor(z, device=dev, requires_grad=True)
    gt = [torch.tensor(g, device=dev, requires_grad=True) for g in geom]
    img_t = T(zt, None, gt, positions=torch.from_numpy(pos).to(dev), noise_mode="const")
    loss_t = ((img_t - torch.from_numpy(target).to(dev)) ** 2).mean()
    assert abs(float(loss_t.detach()) - float(loss_o.detach())) <= 1e-5 * max(1.0, abs(float(loss_o.detach())))
    params = dict(T.named_reference_parameters())
    grads_t = torch.autograd.grad(loss_t, [params[k] for k in train_keys] + [zt] + gt, allow_unused=True)
    checked = 0
    for name, a, b in zip(train_keys + ["z", "geom0", "geom1"], grads_t, grads_o):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, name
            continue
        scale = max(float(b.abs().max()), 1e-8)
        err = float((a.cpu() - b).abs().max())
        assert err <= 2e-4 * scale + 1e-9, (name, err, scale)
        checked += 1
    assert checked >= len(train_keys) - 2


def test_discriminator_matches_reference_golden():
    """TrainableDiscriminator (resnet, c_dim=0) on the HIP operators against the REFERENCE discriminator's logits,
    first-order gradients (image and every parameter) and the R1 double backward (tests/golden/discriminator_r32.npz)."""
    import ast
    from conftest import load_golden
    from brushstroke_engine_amd.training import TrainableDiscriminator
    g = load_golden("discriminator_r32.npz")
    kw = ast.literal_eval(str(g["kw"][0]))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    dev = torch.device("cuda:0")
    D = TrainableDiscriminator(sd, kw["img_resolution"], kw["img_channels"], channel_base=kw["channel_base"],
                               channel_max=kw["channel_max"], conv_clamp=kw["conv_clamp"], device=dev)
    img = torch.tensor(g["img"], device=dev, requires_grad=True)
    logits = D(img, None)
    assert float((logits.detach().cpu() - torch.from_numpy(g["logits"])).abs().max()) <= 2e-5 * max(1.0, float(np.abs(g["logits"]).max()))
    names = [k[2:] for k in g if k.startswith("g.")]
    params = dict(D.named_reference_parameters())
    grads = torch.autograd.grad(logits.sum(), [img] + [params[n] for n in names], create_graph=True)

    def chk(got, want, what, tol=3e-4):
        scale = max(float(np.abs(want).max()), 1e-6)
        err = float((got.detach().cpu() - torch.from_numpy(want)).abs().max())
        assert err <= tol * scale, (what, err, scale)
    chk(grads[0], g["dimg"], "dimg")
    for n, a in zip(names, grads[1:]):
        chk(a, g["g." + n], "g." + n)
    r1 = grads[0].square().sum()
    assert abs(float(r1.detach()) - float(g["r1"][0])) <= 3e-4 * float(g["r1"][0])
    g2 = torch.autograd.grad(r1, [params[n] for n in names], allow_unused=True)
    for n, a in zip(names, g2):
        want = g["r1." + n]
        if a is None:
            assert float(np.abs(want).max()) == 0.0, n
        else:
            chk(a, want, "r1." + n, tol=1e-3)


def test_gan_loss_phases():
    """GanLoss phases accumulate the gradients the formulas of loss_modified.py:140-272 prescribe: Gmain reaches only G,
    the D phases only D; the R1 statistic equals the penalty computed by hand from D's input gradient."""
    from brushstroke_engine_amd.training import (TrainableGenerator, TrainableDiscriminator, GanLoss,
                                                 random_discriminator_state_dict)
    cfg, sd, z, geom, pos = _setup(n=4)
    dev = torch.device("cuda:0")
    G = TrainableGenerator(cfg, sd, dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(32, 3, channel_base=512, channel_max=24, seed=3, bias_std=0.1),
                               32, 3, channel_base=512, channel_max=24, conv_clamp=256, device=dev)
    loss = GanLoss(G, D, r1_gamma=10.0)                      # reference behaviour: random noise + style mixing
    zt = torch.from_numpy(z).to(dev); gt = [torch.from_numpy(g).to(dev) for g in geom]
    real = torch.tanh(torch.randn(4, 3, 32, 32, device=dev))
    w_avg0 = G.p("mapping.w_avg").clone()
    st = loss.accumulate_gradients("Gmain", real, gt, zt)
    assert np.isfinite(st["Loss/G/loss"])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in G.parameters() if p.requires_grad and p.numel() > 1)
    # Gmain back-propagates THROUGH D with D's parameters frozen (the reference loop brackets the phase with requires_grad_)
    assert all(p.grad is None for p in D.parameters()) and all(p.requires_grad for p in D.parameters())
    assert not torch.equal(G.p("mapping.w_avg"), w_avg0)    # the w_avg EMA moves in training mode (networks.py:274-276)
    for p in list(G.parameters()) + list(D.parameters()):
        p.grad = None
    st = loss.accumulate_gradients("Dall", real, gt, zt)
    assert all(p.grad is None for p in G.parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in D.parameters())
    r = real.clone().requires_grad_(True)
    gr, = torch.autograd.grad(D(r, None).sum(), [r])
    assert abs(st["Loss/r1_penalty"] - float(gr.square().sum([1, 2, 3]).mean())) <= 1e-4 * st["Loss/r1_penalty"]


def test_dmain_stacked_batch_equals_two_passes():
    """'Dmain' with generated and real images stacked into one discriminator batch (GanLoss.merge_d_passes, the default) gives the
    gradients and statistics of the reference's two passes (loss_modified.py:223-238): gradients add linearly and the
    minibatch-stddev statistics are taken per half.  Constant noise and no style mixing / augmentation make both runs see the
    same generated images; the discriminator's stacked forward is also checked against separate calls."""
    from brushstroke_engine_amd.training import (TrainableGenerator, TrainableDiscriminator, GanLoss,
                                                 random_discriminator_state_dict)
    cfg, sd, z, geom, pos = _setup(n=8)
    dev = torch.device("cuda:0")
    G = TrainableGenerator(cfg, sd, dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(32, 3, channel_base=512, channel_max=24, seed=3, bias_std=0.1),
                               32, 3, channel_base=512, channel_max=24, conv_clamp=256, device=dev)
    zt = torch.from_numpy(z).to(dev); gt = [torch.from_numpy(g).to(dev) for g in geom]
    real = torch.tanh(torch.randn(8, 3, 32, 32, device=dev))
    with torch.no_grad():
        a, b = torch.randn(8, 3, 32, 32, device=dev), torch.randn(8, 3, 32, 32, device=dev)
        both = D(torch.cat([a, b]), None, sub_batches=2)
        assert float((both - torch.cat([D(a, None), D(b, None)])).abs().max()) <= 1e-5 * float(both.abs().max())
    res = []
    for merged in (True, False):
        loss = GanLoss(G, D, noise_mode="const", style_mixing_prob=0.0, merge_d_passes=merged)
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        st = loss.accumulate_gradients("Dmain", real, gt, zt)
        res.append((dict(st.items()), [p.grad.clone() for p in D.parameters()], loss.real_sign_count, float(loss.real_sign_sum)))
    assert res[0][2:] == res[1][2:]
    for k in ("Loss/D/loss_gen", "Loss/D/loss_real"):
        assert abs(res[0][0][k] - res[1][0][k]) <= 1e-5 * abs(res[1][0][k])
    for ga, gb in zip(res[0][1], res[1][1]):
        assert float((ga - gb).abs().max()) <= 2e-4 * max(1e-8, float(gb.abs().max()))


def test_path_length_regulariser_matches_oracle():
    """'Greg' phase: gradients of the path-length penalty (a second-order quantity: it differentiates d(img . noise)/d ws)
    w.r.t. the generator parameters, HIP autograd path vs the oracle under torch.autograd on CPU, same pl noise."""
    from brushstroke_engine_amd.training import (TrainableGenerator, TrainableDiscriminator, GanLoss,
                                                 random_discriminator_state_dict)
    from oracle import neube_oracle as orc
    cfg, sd, z, geom, pos = _setup(n=4)
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(2)
    b = 2                                                                  # pl_batch_shrink = 2
    noise = (rs.randn(b, 3, cfg.img_resolution, cfg.img_resolution) / cfg.img_resolution).astype(np.float32)
    # oracle in float64 (a second-order quantity through 8 layers: fp32 evaluations of it differ at the 1e-2 level)
    O = orc.OracleGenerator(cfg, sd, dtype=torch.float64)
    keys = [k for k in O.sd if k.endswith((".weight", ".bias", ".noise_strength", ".const", ".color_bias"))]
    for k in keys:
        O.sd[k].requires_grad_(True)
    ws = O.mapping(torch.tensor(z[:b], dtype=torch.float64))
    img, _ = O.synthesis(ws, [torch.tensor(g[:b], dtype=torch.float64) for g in geom], return_debug_data=True)
    plg, = torch.autograd.grad((img * torch.tensor(noise, dtype=torch.float64)).sum(), [ws], create_graph=True)
    pl_len = plg.square().sum(2).mean(1).sqrt()
    pl_mean = torch.zeros([], dtype=torch.float64).lerp(pl_len.mean(), 0.01).detach()
    loss_o = ((pl_len - pl_mean).square() * 2.0).mean()
    grads_o = torch.autograd.grad(loss_o, [O.sd[k] for k in keys], allow_unused=True)
    # HIP
    G = TrainableGenerator(cfg, sd, dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(32, 3, channel_base=512, channel_max=24), 32, 3, channel_base=512,
                               channel_max=24, device=dev)
    loss = GanLoss(G, D, style_mixing_prob=0, noise_mode="const")       # deterministic forward: compared with the oracle
    st = loss.accumulate_gradients("Greg", None, [torch.from_numpy(g).to(dev) for g in geom], torch.from_numpy(z).to(dev),
                                   pl_noise=torch.from_numpy(noise).to(dev))
    want_pen = float((pl_len - pl_mean).square().mean().detach())
    assert abs(st["Loss/pl_penalty"] - want_pen) <= 1e-3 * want_pen
    params = dict(G.named_reference_parameters())
    checked = 0
    for k, go in zip(keys, grads_o):
        gt = params[k].grad
        if go is None or float(go.abs().max()) == 0.0:
            assert gt is None or float(gt.abs().max()) <= 1e-7
            continue
        scale = float(go.abs().max())
        err = float((gt.cpu().double() - go).abs().max())
        assert err <= 2e-2 * scale + 1e-9, (k, err, scale)
        checked += 1
    assert checked >= 20


def test_ddp_gradients_equal_full_batch(tmp_path):
    """Data parallelism of the training path: two ranks (both on cuda:0, gloo backend) wrap the differentiable generator in
    DistributedDataParallel and each back-propagates its half of the batch; the all-reduced (averaged) gradients must
    equal the single-process gradients of the mean loss over the whole batch."""
    import os, socket, subprocess, sys
    from brushstroke_engine_amd.training import TrainableGenerator
    cfg, sd, z, geom, pos = _setup(n=4)
    dev = torch.device("cuda:0")
    G = TrainableGenerator(cfg, sd, dev)
    target = np.random.RandomState(1).randn(4, 3, 32, 32).astype(np.float32)
    img = G(torch.from_numpy(z).to(dev), None, [torch.from_numpy(g).to(dev) for g in geom], noise_mode="const")
    (img - torch.from_numpy(target).to(dev)).square().mean().backward()
    want = {k: p.grad.cpu().numpy() for k, p in G.named_reference_parameters() if p.grad is not None}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "ddp_grads.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ddp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", out], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    assert set(got.files) == set(want)
    for k in want:
        scale = max(float(np.abs(want[k]).max()), 1e-8)
        assert float(np.abs(got[k] - want[k]).max()) <= 1e-4 * scale + 1e-9, k


def test_config5_full_size_iteration():
    """BASELINE config 5 AT ITS SIZE: one full iteration of the StyleGAN2-ADA schedule at R=256, style1 channel widths, batch 8
    -- Gmain + Greg (path length), Dmain + Dreg (R1), Ggeom, ADA 'bgc' pipe in front of D, random noise and style mixing,
    Adam steps -- is finite everywhere, moves every trained parameter, and a second iteration still is."""
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.augment import AugmentPipe
    from brushstroke_engine_amd.training import TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict
    dev, res, n = torch.device("cuda:0"), 256, 8
    torch.manual_seed(0)
    cfg = cfgmod.style1_config(res)
    G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 0), dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(res, 3, channel_base=16384, channel_max=128), res, 3,
                               channel_base=16384, channel_max=128, conv_clamp=256, device=dev)
    pipe = AugmentPipe(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1,
                       saturation=1).to(dev)
    pipe.p.fill_(0.3)
    loss = GanLoss(G, D, augment_pipe=pipe, geom_phase_losses="1.0*iou_inv(uvs)", geom_warmstart_losses="1.0*iou_inv(uvs)+1.0*iou(u)")
    optG = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.0, 0.99))
    optD = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, 0)]
    real = torch.tanh(torch.nn.functional.interpolate(torch.randn(n, 3, 8, 8, device=dev), size=res, mode="bilinear"))
    real_geom = (torch.rand(n, 1, res, res, device=dev) > 0.08).float()
    before = {k: p.detach().clone() for k, p in list(G.named_parameters()) + list(D.named_parameters())}
    finite = lambda m: all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    for it in range(2):
        z = torch.randn(n, cfg.z_dim, device=dev)
        stats = {}
        optG.zero_grad(set_to_none=True)
        stats.update(loss.accumulate_gradients("Gmain", real, geom, z))
        if it == 0:
            stats.update(loss.accumulate_gradients("Greg", real, geom, z, gain=4))
        assert finite(G)
        optG.step()
        optD.zero_grad(set_to_none=True)
        stats.update(loss.accumulate_gradients("Dmain", real, geom, z))
        if it == 0:
            stats.update(loss.accumulate_gradients("Dreg", real, geom, z, gain=16))
        assert finite(D)
        optD.step()
        if it == 0:
            optG.zero_grad(set_to_none=True)
            stats.update(loss.accumulate_gradients("Ggeom", real, geom, z, real_geom=real_geom))
            assert finite(G)
            optG.step()
        assert stats and all(np.isfinite(float(v)) for v in stats.values()), stats
        import json
        assert all(isinstance(v, float) for v in stats.values()) and json.loads(json.dumps(stats)).keys() == stats.keys()      # (merged through dict.update: floats, no device tensors)
        if it == 0:
            assert {"Loss/G/loss", "Loss/pl_penalty", "Loss/r1_penalty"} <= set(stats), sorted(stats)
    moved = [k for k, p in list(G.named_parameters()) + list(D.named_parameters()) if p.requires_grad and not torch.equal(p, before[k])]
    assert len(moved) >= 0.9 * sum(1 for _, p in list(G.named_parameters()) + list(D.named_parameters()) if p.requires_grad)
    assert all(bool(torch.isfinite(p).all()) for p in list(G.parameters()) + list(D.parameters()))


def test_config5_full_size_two_ranks_equal_accumulated_shards(tmp_path):
    """Config 5's data parallelism at its size: two ranks (4 + 4 samples at R=256, both on cuda:0, gloo) run Gmain and Dmain and
    all-reduce the flattened gradients (GanLoss.all_reduce_gradients).  The discriminator's minibatch-stddev layer
    (networks.py:746-769) takes its statistics over the samples a replica sees -- in the reference too, where every GPU
    runs ``batch_gpu`` samples -- so the reduced gradients equal the average of the two shards' gradients evaluated one after
    the other in ONE process (gradient accumulation over the same two half-batches), not those of one batch of 8."""
    import os, socket, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _train_worker as tw
    dev = torch.device("cuda:0")
    G, D, loss, z, geom, real = tw.setup(dev)
    n = z.shape[0]
    parts = []
    for a, b in ((0, n // 2), (n // 2, n)):
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        parts.append(tw.grads(loss, G, D, z, geom, real, dev, a, b))
    wantG, wantD = (parts[0][0] + parts[1][0]) / 2, (parts[0][1] + parts[1][1]) / 2
    assert np.isfinite(wantG).all() and np.isfinite(wantD).all() and np.abs(wantG).max() > 0 and np.abs(wantD).max() > 0
    assert float(np.abs(parts[0][0] - parts[1][0]).max()) > 0.05 * float(np.abs(wantG).max())       # (the shards do differ)
    del G, D, loss
    torch.cuda.empty_cache()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "grads.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_train_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", out], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    for name, want in (("G", wantG), ("D", wantD)):
        assert got[name].shape == want.shape
        assert float(np.abs(got[name] - want).max()) <= 2e-4 * float(np.abs(want).max()), name


def test_forger_geometry_and_stitch_phases():
    """Ggeom / Ggeom-warm / Gstitch of ForgerLoss (loss_modified.py:108-138, 181-203) on the differentiable HIP generator
    and discriminator: the geometry phase's loss equals the loss items evaluated on the CPU oracle's uvs, its gradients
    equal oracle autograd, only G receives gradients; the stitch phase composites two overlapping crops and trains G
    through a frozen D."""
    from brushstroke_engine_amd.training import (TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict)
    from brushstroke_engine_amd import forger_losses as fl
    from oracle import neube_oracle as orc
    cfg, sd, z, geom, pos = _setup(n=2)
    dev = torch.device("cuda:0")
    G = TrainableGenerator(cfg, sd, dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(32, 3, channel_base=512, channel_max=24, seed=3, bias_std=0.1),
                               32, 3, channel_base=512, channel_max=24, conv_clamp=256, device=dev)
    loss = GanLoss(G, D, geom_phase_losses="1.0*iou_inv(uvs)", geom_warmstart_losses="1.0*iou_inv(uvs)+1.0*iou(u)",
                   stitch_phase_losses="gan(fake_composite)+0.5*l1(patch)", main_phase_losses="0.1*iou(u)",
                   style_mixing_prob=0, noise_mode="const", stitcher=fl.RandomStitcher(crop_margin=2, min_overlap=8))
    zt = torch.from_numpy(z).to(dev); gt = [torch.from_numpy(g).to(dev) for g in geom]
    pt = torch.from_numpy(pos).to(dev)
    truth = (torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(1)) > 0.3).float()
    # oracle under autograd: the same phase on CPU
    O = orc.OracleGenerator(cfg, sd)
    keys = [k for k in O.sd if k.endswith((".weight", ".bias", ".const", ".noise_strength", ".color_bias"))]
    for k in keys:
        O.sd[k].requires_grad_(True)
    npos = ((pos % cfg.img_resolution) / (cfg.img_resolution - 1)).astype(np.float32)
    _, dbg = O.synthesis(O.mapping(torch.tensor(z)), [torch.tensor(g_) for g_ in geom], return_debug_data=True, norm_noise_positions=npos)
    want, _ = fl.ForgerLosses.create_from_string("1.0*iou_inv(uvs)+1.0*iou(u)").compute({"uvs": dbg["uvs"]}, truth)
    grads_o = torch.autograd.grad(want, [O.sd[k] for k in keys], allow_unused=True)
    st = loss.accumulate_gradients("Ggeom-warm", None, gt, zt, positions=pt, real_geom=truth.to(dev))
    got = st["Loss/forger/Ggeom-warm/iou_inv_uvs"] + st["Loss/forger/Ggeom-warm/iou_u"]
    assert abs(got - float(want.detach())) <= 1e-4 * abs(float(want.detach()))
    params = dict(G.named_reference_parameters())
    checked, worst = 0, {}
    for k, go in zip(keys, grads_o):
        gp = params[k].grad
        if go is None or float(go.abs().max()) < 1e-12:
            continue
        assert gp is not None, k
        worst[k] = float((gp.cpu() - go).abs().max()) / float(go.abs().max())
        checked += 1
    bad = {k: round(v, 5) for k, v in worst.items() if v > 1e-3}
    assert not bad, bad
    assert checked >= 20 and all(p.grad is None for p in D.parameters())
    for p in G.parameters():
        p.grad = None
    st = loss.accumulate_gradients("Ggeom", None, gt, zt, positions=pt, real_geom=truth.to(dev))
    assert set(st) == {"Loss/forger/Ggeom/iou_inv_uvs"} and np.isfinite(list(st.values())[0])
    st = loss.accumulate_gradients("Gmain", None, gt, zt, positions=pt, real_geom=truth.to(dev))
    assert "Loss/forger/Gmain/iou_u" in st and np.isfinite(st["Loss/G/loss"])
    for p in G.parameters():
        p.grad = None
    # stitch phase: two overlapping crops (their geometry features stand for the two crops of one drawing)
    g2 = [torch.from_numpy(a).to(dev) for a in synthetic.geom_features(cfg, 2, 8)]
    st = loss.accumulate_gradients_stitch(gt, g2, (40, 44, 32, 32), (48, 38, 32, 32), zt, gain=1.0, positions1=pt)
    assert np.isfinite(st["Loss/forger/Gstitch/total"]) and {"Loss/forger/Gstitch/gan_fake_composite", "Loss/forger/Gstitch/l1_patch"} <= set(st)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in G.parameters() if p.requires_grad and p.numel() > 1)
    assert all(p.grad is None for p in D.parameters()) and all(p.requires_grad for p in D.parameters())
    with pytest.raises(RuntimeError, match="LPIPS"):
        GanLoss(G, D, stitch_phase_losses="lpips(patch)")
