"""Host issue time per step of the throughput schedules against the device time (is the step loop host-bound?):  python tools/host_issue_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
from brushstroke_engine_amd.pipeline import ConcurrentTriadSteps

dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, seed=0)).to(dev)
B = 32
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev).float()
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, seed=0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, seed=0)).to(dev)
for k in (1, 3):
    s = ConcurrentTriadSteps(G, streams=k)
    for _ in range(30):
        s.submit(z, geom, pos)
    s.wait(); torch.cuda.synchronize()
    for rep in range(3):
        n = 60
        t0 = time.perf_counter()
        for _ in range(n):
            s.submit(z, geom, pos)
        t1 = time.perf_counter()
        s.wait(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"streams {k}: host issue {(t1 - t0) / n * 1e3:.3f} ms/step, total {(t2 - t0) / n * 1e3:.3f} ms/step ({B * n / (t2 - t0):.0f} patches/s); host share {(t1 - t0) / (t2 - t0):.2f}")
