"""Same-box A/B: the layers that compute their noise themselves normalise the batch's integer positions at the top of every tile
(SynthesisNetwork.positions_once = False: four 64-bit modulo operations per lane and tile) or read positions normalised once per batch
by nb_norm_positions_f32 (True, default); beside both, the caller-normalised form (norm_noise_positions).  Synthesis passes of batch 32
at R=256 on one stream, alternating; outputs compared bit for bit.      gpurun -- 'python tools/ab_positions_once.py'"""
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


def run(once, reps=30):
    G.synthesis.positions_once = once
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
    print(f"per tile {t0:.4f} ms   once per batch {t1:.4f} ms   ({(t0 / t1 - 1) * 100:+.1f} %)   images bit-identical")
G.synthesis.positions_once = True
