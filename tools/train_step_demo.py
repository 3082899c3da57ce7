"""Row f4 demo: a few optimiser steps through the differentiable generator (brushstroke_engine_amd.training) - fit the
image of one latent to the image of another by gradient descent on all generator parameters - with step timings.
Forward and backward of every modulated conv / bias_act / upfirdn2d run on the HIP kernels (first, untuned versions)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.training import TrainableGenerator
res, n = int(os.environ.get("NB_RES", "128")), int(os.environ.get("NB_B", "4"))
cfg = cfgmod.style1_config(res)
dev = torch.device("cuda:0")
T = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 0), dev)
z = torch.from_numpy(synthetic.batch_z(cfg, n, 0)).float().to(dev)
z2 = torch.from_numpy(synthetic.batch_z(cfg, n, 100)).float().to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, 0)]
with torch.no_grad():
    target = T(z2, None, geom)
opt = torch.optim.Adam(T.parameters(), lr=2e-3)
losses, times = [], []
for it in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = T(z, None, geom)
    loss = (img - target).square().mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    losses.append(float(loss.detach()))
print(f"R={res} batch {n}: loss {losses[0]:.5f} -> {losses[-1]:.5f} in {len(losses)} Adam steps; "
      f"forward+backward+step {np.median(times[2:]):.1f} ms (median)")
assert losses[-1] < losses[0]
