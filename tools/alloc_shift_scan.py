"""Developer tool: does the step time depend on WHERE the allocator places the generator's buffers?  Runs the bench's steady-state
loop after a dummy allocation of NB_SHIFT_MB megabytes (which moves every later hipMalloc) and prints ms/step -- call it in a
shell loop over sizes (each run is a fresh process, as the placement is fixed for a process's lifetime)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
mb = float(os.environ.get("NB_SHIFT_MB", "0"))
dummy = torch.empty([int(mb * (1 << 20))], dtype=torch.uint8, device=dev) if mb > 0 else None
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode="f8").to(dev)
G.sub_streams = int(os.environ.get("NB_SUB", "1"))
B = 32
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
for _ in range(20): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print(f"shift {mb:8.2f} MB: {dt * 1e3:.3f} ms/step")
