"""Batch-1 hipGraph latency for different split-f16 thresholds (h3_min_batch) and small batches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
from brushstroke_engine_amd.graphed import GraphedTriadRender
dev = torch.device("cuda:0")
for res in (256, 128):
    cfg = cfgmod.style1_config(res)
    G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode=os.environ.get("NB_MODE", "h3")).to(dev)
    for B in (1, 2, 4):
        for minb in (1, 4, 64):
            G.synthesis.h3_min_batch = minb
            G._invalidate()
            z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
            geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
            pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
            gr = GraphedTriadRender(G, batch=B)
            gr.set_inputs(z=z, geom_feature=geom, positions=pos)
            for _ in range(20): gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            print(f"R={res} batch {B} h3_min_batch {minb:2d} ({'h3' if B >= minb else 'f32'}): p50 {np.percentile(ts, 50):.3f} ms  p99 {np.percentile(ts, 99):.3f} ms")
