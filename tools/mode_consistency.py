"""Cross-mode consistency over resolutions and batch sizes (incl. odd batches that split unevenly over the sub-batch
streams): f8 / h3 against the fp32-MFMA mode, whole generator."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
worst = {"h3": 0.0, "f8": 0.0}
for res in (64, 128, 256, 512):
    cfg = cfgmod.style1_config(res); sd = wmod.random_state_dict(cfg, 1)
    Gs = {m: Generator(cfg, sd, conv_mode=m).to(dev) for m in ("f32", "h3", "f8")}
    for n in (1, 3, 16, 33) if res <= 256 else (1, 5, 17):
        z = torch.from_numpy(synthetic.batch_z(cfg, n, 7)).to(dev)
        geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, 3)]
        pos = torch.from_numpy(synthetic.positions(cfg, n, 3)).to(dev)
        ref = Gs["f32"].render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
        line = f"R={res} n={n}:"
        for m in ("h3", "f8"):
            out = Gs[m].render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
            e = float((out[1] - ref[1]).abs().max()); eu = float((out[2]["uvs"] - ref[2]["uvs"]).abs().max())
            d8 = int((out[0].int() - ref[0].int()).abs().max())
            worst[m] = max(worst[m], e, eu)
            line += f"  {m}: rgba {e:.1e} uvs {eu:.1e} u8 max {d8}"
            assert d8 <= 1 and e < (3e-4 if m == "f8" else 5e-5), (m, res, n, e)
        print(line)
print("worst", worst)
