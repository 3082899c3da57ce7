"""Run-to-run determinism of the whole batch-32 step (NB_SUB sub-streams): 12 runs, mismatching output bytes vs the first."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
cfg = cfgmod.style1_config(256)
for mode in os.environ.get("NB_MODES", "f8 h3").split():
    G = Generator(cfg, wmod.random_state_dict(cfg, 2), conv_mode=mode).to(dev)
    G.sub_stream_min_batch = 16                     # two sub-batch chains also at batch 32 (the default starts them at 64)
    if os.environ.get("NB_SUB"):
        G.sub_streams = int(os.environ["NB_SUB"])
    B = 32
    z = torch.from_numpy(synthetic.batch_z(cfg, B, 40)).to(dev)
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, B, 4)]
    pos = torch.from_numpy(synthetic.positions(cfg, B, 4)).to(dev)
    outs = []
    for _ in range(12):
        u8, rgba, dbg = G.render_triad(z=z, geom_feature=geom, positions=pos, want_f32=True)
        torch.cuda.synchronize()
        outs.append((u8.clone(), dbg["uvs"].clone()))
    print(mode, "sub_streams", G.sub_streams, "mismatching uvs elements vs run 0:", [int((outs[0][1] != o[1]).sum()) for o in outs[1:]])
