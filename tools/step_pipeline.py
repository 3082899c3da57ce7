"""Throughput of the batch-32 step enqueued on ONE stream vs alternating whole steps between TWO streams (separate workspaces):
the second chain's kernels fill the CUs that one chain leaves idle at its small / latency-bound launches and kernel tails."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator

dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
mode = os.environ.get("NB_MODE", "f8")
G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode=mode).to(dev)
B = int(os.environ.get("NB_BATCH", "32"))
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, seed=0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, seed=0)).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(3)]


def run(nstreams, steps=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if nstreams == 1:
            G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
        else:
            k = i % nstreams
            with torch.cuda.stream(streams[k]):
                G.render_triad(z=z, geom_feature=geom, positions=pos, join=False, _plan_slot=20 + k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for ns in (1, 2, 3, 1, 2, 3):
    run(ns, 10)
    ms = run(ns)
    print(f"{mode} batch {B}: {ns} stream(s): {ms:.3f} ms/step = {B / ms * 1e3:.0f} patches/s")
