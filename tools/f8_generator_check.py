"""f8 conv mode end to end: error vs the h3 and f32 modes, and throughput."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
for res in (128, 256):
    cfg = cfgmod.style1_config(res); sd = wmod.random_state_dict(cfg, 0)
    B = 32
    z = torch.from_numpy(synthetic.batch_z(cfg, B, 0)).to(dev)
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 0)]
    pos = torch.from_numpy(synthetic.positions(cfg, B, 0)).to(dev)
    outs = {}
    for mode in ("f32", "h3", "f8"):
        G = Generator(cfg, sd, conv_mode=mode).to(dev)
        u8, f32, dbg = G.render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
        for _ in range(10): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): G.render_triad(z=z, geom_feature=geom, positions=pos, join=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        outs[mode] = (u8, f32, dbg["uvs"], dt)
        print(f"R={res} {mode}: {dt * 1e3:.3f} ms/step {B / dt:.0f} patches/s; kernels {sorted(set(G.synthesis.layer_kernels.values()))[:3]}")
    for mode in ("h3", "f8"):
        print(f"   {mode} vs f32: rgba max abs diff {float((outs[mode][1] - outs['f32'][1]).abs().max()):.2e}, uvs {float((outs[mode][2] - outs['f32'][2]).abs().max()):.2e},"
              f" u8 bytes differing {float((outs[mode][0] != outs['f32'][0]).float().mean()) * 100:.3f} %")
