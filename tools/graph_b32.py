import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
from brushstroke_engine_amd.graphed import GraphedTriadRender
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256); sd = wmod.random_state_dict(cfg, 0)
G = Generator(cfg, sd).to(dev)
B = 32
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
for _ in range(3): G.render_triad(z=z, geom_feature=geom, positions=pos)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): G.render_triad(z=z, geom_feature=geom, positions=pos)
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3)
gr = GraphedTriadRender(G, batch=B)
gr.set_inputs(z=z, geom_feature=geom, positions=pos)
for _ in range(3): gr.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): gr.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3)
# host-side enqueue time of one eager step (no GPU wait): is the eager path host-bound?
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G.render_triad(z=z, geom_feature=geom, positions=pos)
    ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
print("host enqueue ms/step (eager)", sorted(ts)[len(ts) // 2])
