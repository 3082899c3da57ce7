"""Per-launch durations of the batch-1 step (eager, HIP events around every launch) for the conv modes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256); sd = wmod.random_state_dict(cfg, 0)
B = int(os.environ.get("NB_B", "1"))
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
for mode, minb in (("f32", 99), ("h3", 1), ("f8", 1)):
    G = Generator(cfg, sd, conv_mode=mode).to(dev)
    G.synthesis.h3_min_batch = minb
    for _ in range(5): G.render_triad(z=z, geom_feature=geom, positions=pos)
    G.synthesis.layer_events = []
    for _ in range(5): G.render_triad(z=z, geom_feature=geom, positions=pos)
    torch.cuda.synchronize()
    acc = {}
    for name, e0, e1 in G.synthesis.layer_events: acc.setdefault(name, []).append(e0.elapsed_time(e1))
    G.synthesis.layer_events = None
    tot = sum(np.mean(v) for v in acc.values())
    print(mode, f"sum of launches {tot * 1e3:.0f} us:", {k.replace('synthesis.', ''): round(float(np.mean(v)) * 1e3) for k, v in acc.items()})
