"""Developer aid (GPU box): run the HIP generator layer by layer next to the CPU oracle and print
the max-abs error after every layer.  usage: python tools/debug_layers.py [tiny|64|128|256] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic, ops, _lib
from brushstroke_engine_amd.networks import Generator
from oracle import neube_oracle as orc

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = cfgmod.tiny_config(32) if which == "tiny" else cfgmod.style1_config(int(which))
sd = wmod.random_state_dict(cfg, seed=11)
dev = torch.device("cuda:0")
G = Generator(cfg, sd).to(dev)
O = orc.OracleGenerator(cfg, sd)
z = synthetic.batch_z(cfg, n, 594)
geom = synthetic.geom_features(cfg, n, seed=3)
pos = synthetic.positions(cfg, n, seed=1)
if os.environ.get("POS"):
    pos = np.array(eval(os.environ["POS"]), np.int64)[:n]
taps = {}
want_img, want = O(z, None, geom, positions=pos, return_debug_data=True, taps=taps)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ws = G.mapping(D(z))
print("ws", float((ws.cpu() - want["ws"]).abs().max()))
syn = G.synthesis
feats = [r for r in cfg.block_resolutions]
img, dbg = G.forward_pre_mapped(ws, [D(g) for g in geom], positions=D(pos), return_debug_data=True,
                                return_features=feats, noise_mode="const", _extra_outputs=(ex := {"logits": True}))
plan = syn._plans[0]
for i, s in enumerate(cfg.layers):
    st = plan.styles[i][:n].cpu()
    npos = (torch.from_numpy(pos) % cfg.img_resolution) / (cfg.img_resolution - 1)
    wn = orc.shifted_const_noise(O.sd[s.name + ".noise_const"], O.sd[s.name + ".noise_grid"], npos)[:, 0] * O.sd[s.name + ".noise_strength"]
    print(f"noise err {float((plan.noise[i][:n].cpu() - wn).abs().max()):.2e}", end=" ")
    print(f"{s.name:28s} styles err {float((st - taps[s.name + '.styles']).abs().max()):.2e}", end="  ")
    if s.up == 1:
        got = dbg[f"features{s.block_res}"].cpu()
        print(f"out err {float((got - taps[s.name + '.out']).abs().max()):.2e}")
    else:
        print()
print("logits", float((ex["out"]["logits"].cpu() - taps["torgb.logits"]).abs().max()))
print("uvs", float((dbg["uvs"].cpu() - want["uvs"]).abs().max()), "img", float((img.cpu() - want_img).abs().max()))
