"""BASELINE config 3 on its named input, one GPU: lamali_sm.png (514 x 800; its thresholded geometry ships in
tests/golden/engine_lamali_r256.npz) through PaintingHelper.paint_image at P = 256, crop margin 10, feature blending
level 2 = 12 tiles -- wall-clock from the host geometry array to the host RGBA canvas, and the distance of the canvas
from the one the REFERENCE engine painted (the fixture)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
from brushstroke_engine_amd.networks import Generator

g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "engine_lamali_r256.npz")))
h, w = g["geom_shape"].tolist()
geom = (np.unpackbits(g["geom_bits"])[:h * w].reshape(h, w) * 255).astype(np.uint8)
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, seed=int(g["weights_seed"]))).to("cuda")
ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(int(g["encoder_seed"]))))
z = np.random.RandomState(int(g["style_seed"])).randn(1, cfg.z_dim)
res = {}
for level in (2, 0):
    helper = painting.PaintingHelper(ops, batch=32)
    helper.set_feature_blending(level)
    opts = painting.GanBrushOptions()
    opts.set_style(torch.from_numpy(z), 594)
    for _ in range(3):
        out, full, crops, padded = helper.paint_image(geom, opts, crop_margin=10, return_full=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        out = helper.paint_image(geom, opts, crop_margin=10)
        ts.append(time.perf_counter() - t0)
    res[level] = {"ms_p50": round(float(np.percentile(ts, 50)) * 1e3, 3), "tiles_per_s": round(len(crops) / float(np.percentile(ts, 50)), 1)}
    if level == 2:
        d = np.abs(full.astype(np.int32) - g["canvas_level2_clear"].astype(np.int32))
        res[level]["vs_reference_canvas"] = {"max_lsb": int(d.max()), "bytes_differing": float((d > 0).mean())}
print(json.dumps({"metric": "lamali_sm.png end to end (host geometry -> host RGBA), 12 tiles of 256x256, 1 GPU", "conv_mode": G.synthesis.conv_mode,
                  "feature_blending_2": res[2], "feature_blending_0": res[0], "image": [h, w]}))
