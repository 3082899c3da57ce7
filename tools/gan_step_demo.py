"""Row f4 demo: alternating G / D updates (non-saturating logistic loss + lazy R1, loss_modified.py:140-272) of the
differentiable generator against the resnet discriminator, everything on the HIP operators; prints losses and step times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.training import TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict
res, n = int(os.environ.get("NB_RES", "64")), int(os.environ.get("NB_B", "8"))
cfg = cfgmod.style1_config(res) if res >= 128 else cfgmod.tiny_config(res)
dev = torch.device("cuda:0")
G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 0), dev)
D = TrainableDiscriminator(random_discriminator_state_dict(res, 3, channel_base=16384 if res >= 128 else res * 16, channel_max=128 if res >= 128 else 32),
                           res, 3, channel_base=16384 if res >= 128 else res * 16, channel_max=128 if res >= 128 else 32, conv_clamp=256, device=dev)
loss = GanLoss(G, D, r1_gamma=10.0)
optG = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.0, 0.99)); optD = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
rs = np.random.RandomState(0)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, 0)]
# "real" data: smooth random blobs in [-1, 1]
real = torch.tanh(torch.nn.functional.interpolate(torch.randn(n, 3, 8, 8, device=dev), size=res, mode="bilinear"))
times = []
for it in range(10):
    z = torch.randn(n, cfg.z_dim, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    optG.zero_grad(set_to_none=True); sg = loss.accumulate_gradients("Gmain", real, geom, z); optG.step()
    optD.zero_grad(set_to_none=True); sd = loss.accumulate_gradients("Dmain", real, geom, z)
    if it % 2 == 0: sd.update(loss.accumulate_gradients("Dreg", real, geom, z, gain=2))
    optD.step()
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    print(it, {k: round(v, 4) for k, v in {**sg, **sd}.items()})
    assert all(np.isfinite(v) for v in {**sg, **sd}.values())
print(f"R={res} batch {n}: G step + D step (+ R1 every other) {np.median(times[2:]):.1f} ms (median)")
