import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
sd = wmod.random_state_dict(cfg, 2)
n = 32
z = torch.from_numpy(synthetic.batch_z(cfg, n, 40)).to(dev)
geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, 4)]
pos = torch.from_numpy(synthetic.positions(cfg, n, 4)).to(dev)
pre = os.environ.get("NB_PRE", "f32 h3").split()
for mode in pre + ["f8"]:
    G = Generator(cfg, sd, conv_mode=mode).to(dev)
    if os.environ.get("NB_SUB"): G.sub_streams = int(os.environ["NB_SUB"])
    res = []
    for k in range(4):
        u8, rgba, dbg = G.render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True, return_features=[128] if os.environ.get("NB_FEAT") else None)
        res.append((u8.clone(), rgba.clone(), dbg["uvs"].clone(), dbg.get("features128")))
    torch.cuda.synchronize()
    for k in range(1, 4):
        d = (res[0][1] != res[k][1])
        per_sample = d.flatten(1).sum(1).tolist()
        print(mode, f"call 0 vs {k}: rgba mismatches {int(d.sum())}, uvs {int((res[0][2] != res[k][2]).sum())}, per sample {per_sample if d.any() else ''}",
              "" if res[0][3] is None else f"features128 {int((res[0][3] != res[k][3]).sum())}")
    if mode == "f8" and (res[0][1] != res[1][1]).any():
        d = (res[0][1] != res[1][1])
        idx = d.nonzero()
        print("first mismatches (n, c, y, x):", idx[:8].tolist(), "values", res[0][1][d][:4].tolist(), res[1][1][d][:4].tolist(),
              "max abs diff", float((res[0][1] - res[1][1]).abs().max()))
        ys = idx[:, 2]; xs = idx[:, 3]
        print("y range", int(ys.min()), int(ys.max()), "x range", int(xs.min()), int(xs.max()), "count", idx.shape[0])
