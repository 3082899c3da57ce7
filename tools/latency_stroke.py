"""Interactive stroke latency: PaintingHelper.render_stroke (one 256x256 tile per call, feature canvas carried along),
wall clock from the uint8 stroke patch on the host to the uint8 RGBA tile on the host."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
from brushstroke_engine_amd.networks import Generator
for res in (256, 128):
    cfg = cfgmod.style1_config(res)
    G = Generator(cfg, wmod.random_state_dict(cfg, 0)).to("cuda")
    ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5)))
    for level in (0, 2):
        helper = painting.PaintingHelper(ops)
        helper.make_new_canvas(2048, 2048, feature_blending=level)
        opts = painting.GanBrushOptions()
        opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
        rs = np.random.RandomState(0)
        ts = []
        for i in range(120):
            patch = np.zeros((res, res, 4), np.uint8)
            y0 = rs.randint(10, res - 30)
            patch[y0:y0 + 12, 10:res - 10, 3] = 255
            x, y = int(rs.randint(0, 2048 - res)), int(rs.randint(0, 2048 - res))
            opts.set_position(x, y)
            t0 = time.perf_counter()
            img, _, meta = helper.render_stroke(patch, None, opts, meta={"x": x, "y": y, "crop_margin": 10})
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = ts[20:]
        print(f"R={res} feature blending {level}: render_stroke p50 {np.percentile(ts, 50):.2f} ms  p99 {np.percentile(ts, 99):.2f} ms  out {img.shape}")
