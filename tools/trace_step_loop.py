"""The bench's steady-state loop alone (batch 32, two sub-batch streams, no join), for kernel traces."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode=os.environ.get("NB_MODE", "f8")).to(dev)
B = int(os.environ.get("NB_B", "32"))
if os.environ.get("NB_SUB"):
    G.sub_streams = int(os.environ["NB_SUB"])
if os.environ.get("NB_H3_MIN_PIX"):
    G.synthesis.h3_min_pixels = int(os.environ["NB_H3_MIN_PIX"])
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev).to(torch.float32)       # (as bench.py: no per-step cast)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
for _ in range(60): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(os.environ.get("NB_STEPS", "150"))
for _ in range(N): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{B * N / dt:.0f} patches/s, {dt / N * 1e3:.3f} ms/step")
