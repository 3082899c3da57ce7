"""Developer aid (GPU box): how does the CPU oracle scale with torch threads on this host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from oracle import neube_oracle as orc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
cfg = cfgmod.style1_config(256)
sd = wmod.random_state_dict(cfg, 0)
O = orc.OracleGenerator(cfg, sd)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
z = synthetic.batch_z(cfg, n, 0); geom = synthetic.geom_features(cfg, n, 0); pos = synthetic.positions(cfg, n, 0)
for fused in (True, False):
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        O(z[:1], None, [g[:1] for g in geom], positions=pos[:1], fused_modconv=fused)
        t0 = time.perf_counter(); O(z, None, geom, positions=pos, fused_modconv=fused); dt = time.perf_counter() - t0
        print(f"fused={fused} threads={th} {n/dt:.2f} patches/s ({dt:.2f}s)", flush=True)
        if dt > 40: break
