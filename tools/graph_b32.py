"""Batch-32 step captured into ONE hipGraph (both sub-batch streams inside) vs the eager no-join loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
from brushstroke_engine_amd.graphed import GraphedTriadRender
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, 0), conv_mode=os.environ.get("NB_MODE", "f8")).to(dev)
B = int(os.environ.get("NB_B", "32"))
z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
N = 150
for _ in range(60): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"eager no-join: {B * N / dt:.0f} patches/s, {dt / N * 1e3:.3f} ms/step")
ref = G.render_triad(z=z, geom_feature=geom, positions=pos)[0].clone()
G.sub_stream_min_batch = 10 ** 9                # (a capture runs unsplit: one chain per graph)
gr = GraphedTriadRender(G, batch=B)
gr.set_inputs(z=z, geom_feature=geom, positions=pos)
for _ in range(20): gr.replay()
torch.cuda.synchronize()
assert torch.equal(gr.out_u8, ref), "graph replay differs from eager"
t0 = time.perf_counter()
for _ in range(N): gr.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"one graph per step: {B * N / dt:.0f} patches/s, {dt / N * 1e3:.3f} ms/step")
# P graphs of 1/P of the batch each, replayed on P streams (own workspace each)
for P in (2, 3, 4):
    bounds = [(i * B // P, (i + 1) * B // P) for i in range(P)]
    sts = [torch.cuda.Stream() for _ in range(P)]
    grs = []
    for k, (st, (a, b)) in enumerate(zip(sts, bounds)):
        with torch.cuda.stream(st):
            g_ = GraphedTriadRender(G, batch=b - a, plan_slot=k + 1)
            g_.set_inputs(z=z[a:b], geom_feature=[t_[a:b] for t_ in geom], positions=pos[a:b])
        grs.append(g_)
    torch.cuda.synchronize()
    for _ in range(20):
        for g_, st in zip(grs, sts):
            with torch.cuda.stream(st): g_.replay()
    torch.cuda.synchronize()
    d8 = int((torch.cat([g_.out_u8 for g_ in grs]).int() - ref.int()).abs().max())
    assert d8 <= 1, "part graphs differ from eager"      # (smaller parts move layers between the split-f16 and fp32 kernels: <= 1 LSB)
    t0 = time.perf_counter()
    for _ in range(N):
        for g_, st in zip(grs, sts):
            with torch.cuda.stream(st): g_.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{P} part graphs on {P} streams: {B * N / dt:.0f} patches/s, {dt / N * 1e3:.3f} ms/step")
    del grs
