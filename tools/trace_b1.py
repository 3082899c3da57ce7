"""Batch-1 hipGraph replay loop for a kernel trace (rocprofv3 --kernel-trace): 200 replays of the R=256 step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
from brushstroke_engine_amd.networks import Generator
from brushstroke_engine_amd.graphed import GraphedTriadRender
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode=os.environ.get("NB_MODE", "f8")).to(dev)
if os.environ.get("NB_H3_MIN_PIX"):
    G.synthesis.h3_min_pixels = int(os.environ["NB_H3_MIN_PIX"])
z = torch.from_numpy(synthetic.batch_z(cfg, 1, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, 1, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, 1, 0)).to(dev)
gr = GraphedTriadRender(G, batch=1)
gr.set_inputs(z=z, geom_feature=geom, positions=pos)
for _ in range(20): gr.replay()
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"p50 {np.percentile(ts, 50):.3f} ms  p99 {np.percentile(ts, 99):.3f} ms")
