"""Developer tool: the race behind tools/canvas_race_hunt.py at generator level -- two batches on two streams, lazy encoder, fresh
workspaces (set_conv_mode before every round) -- with every intermediate the generator can return compared against a
single-stream reference, to see which tensor goes wrong first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
from brushstroke_engine_amd.networks import Generator
g = load_golden("engine_r128.npz")
cfg = cfgmod.style1_config(128); sd = wmod.random_state_dict(cfg, seed=0); esd = encmod.random_encoder_state_dict(5)
z = np.random.RandomState(594).randn(1, cfg.z_dim)
dev = torch.device("cuda:0")
G = Generator(cfg, sd, conv_mode="f32").to(dev); enc = encmod.HipGeometryEncoder(esd)
ops = painting.TileOps(G, enc)
ws1 = G.mapping(torch.from_numpy(z).to(dev).float(), None)
geom_dev = ops.to_device(g["geom_padded"])
yx = ops.to_device(g["crops"].astype(np.int32))
tiles = [ops.geom_tiles(geom_dev, yx[0:4]), ops.geom_tiles(geom_dev, yx[4:8])]
pos = [yx[0:4].to(torch.int64), yx[4:8].to(torch.int64)]
RES = [r for r in cfg.block_resolutions if r >= 8]
def run(k, lazy, slot):
    gf = enc.lazy(tiles[k]) if lazy else enc.encode(tiles[k])
    img, dbg = G.forward_pre_mapped(ws1.expand(4, -1, -1).contiguous(), gf, positions=pos[k], noise_mode="const", return_debug_data=True,
                                    return_features=RES, _plan_slot=slot)
    return dict(img=img, uvs=dbg["uvs"], **{f"f{r}": dbg["features"][r] for r in RES if "features" in dbg and r in dbg["features"]})
torch.cuda.synchronize()
ref = [run(0, False, 20), run(1, False, 21)]
torch.cuda.synchronize()
print("reference keys", sorted(ref[0]))
st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
bad = 0
for it in range(int(os.environ.get("NB_ROUNDS", "150"))):
    G.set_conv_mode("f32")
    outs = []
    for k in (0, 1):
        st[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st[k]):
            outs.append(run(k, os.environ.get("NB_EAGER") is None, 8 + k))
    torch.cuda.synchronize()
    for k in (0, 1):
        diffs = {key: float((outs[k][key] - ref[k][key]).abs().max()) for key in ref[k]}
        if any(v > 0 for v in diffs.values()):
            bad += 1
            first = [key for key in sorted(diffs, key=lambda s: (s != "img" and s != "uvs", int(s[1:]) if s[0] == "f" else 10 ** 6)) if diffs[key] > 0]
            where = {}
            for key in first[:3]:
                d = (outs[k][key] - ref[k][key]).abs()
                idx = torch.nonzero(d > 0)
                where[key] = (int(idx.shape[0]), idx[:3].tolist())
            print("round", it, "batch", k, {kk: vv for kk, vv in diffs.items() if vv > 0}, where)
print("rounds with a difference:", bad)
