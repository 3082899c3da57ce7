"""Same-box A/B: the large layers compute their position-shifted noise themselves (SynthesisNetwork.noise_in_kernel = True, default since
round 3) or read the [n, res, res] images the noise launch writes (False).  Synthesis passes of batch 32 at R=256 on one stream,
alternating; outputs compared bit for bit.      gpurun -- 'python tools/ab_noise_in_kernel.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, synthetic, weights as wmod  # noqa: E402
from brushstroke_engine_amd.networks import Generator  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, seed=0)).to(dev)
n = 32
z = torch.from_numpy(synthetic.batch_z(cfg, n, 1)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, seed=0)]
pos = torch.from_numpy(synthetic.positions(cfg, n, seed=0)).to(dev)
_, dbg = G(z, None, geom, positions=pos, return_debug_data=True, noise_mode="const")
ws = dbg["ws"]


def run(inker, reps=30):
    G.synthesis.noise_in_kernel = inker
    for _ in range(5):
        img = G.synthesis(ws, geom, noise_mode="const", _positions=pos)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        img = G.synthesis(ws, geom, noise_mode="const", _positions=pos)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, img.clone()


for i in range(4):
    (t0, a), (t1, b) = run(False), run(True)
    assert torch.equal(a, b)
    print(f"noise images {t0:.4f} ms   in the kernels {t1:.4f} ms   ({(t0 / t1 - 1) * 100:+.1f} %)   images bit-identical")
G.synthesis.noise_in_kernel = True
