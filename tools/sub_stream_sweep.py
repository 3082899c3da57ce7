"""Throughput of the no-join loop for different numbers of sub-batch streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256); sd = wmod.random_state_dict(cfg, 0)
G = Generator(cfg, sd).to(dev)
for B in (32, 64):
    z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
    pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
    for sub in (1, 2, 3, 4):
        G.sub_streams = sub; G._side_streams = None; G.sub_stream_min_batch = 8
        for _ in range(10): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        print(f"batch {B} sub_streams {sub}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} patches/s")
