"""Does splitting the batch over two HIP streams (two half-batches in flight) hide kernel tails? (experiment)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256); sd = wmod.random_state_dict(cfg, 0)
B = 32
def inputs(n, off):
    return (torch.from_numpy(synthetic.batch_z(cfg, n, off)).to(dev), [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, off)],
            torch.from_numpy(synthetic.positions(cfg, n, off)).to(dev))
G = Generator(cfg, sd).to(dev)
z, geom, pos = inputs(B, 0)
def run1():
    G.render_triad(z=z, geom_feature=geom, positions=pos)
for nsplit in (2, 4):
    Gs = [Generator(cfg, sd).to(dev) for _ in range(nsplit)]
    ins = [inputs(B // nsplit, i) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def runs():
        for g, (zz, gg, pp), s in zip(Gs, ins, streams):
            with torch.cuda.stream(s):
                g.render_triad(z=zz, geom_feature=gg, positions=pp)
    for fn, name in ((run1, "1 stream, batch 32"), (runs, f"{nsplit} streams x batch {B // nsplit}")):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"{name}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} patches/s")
