"""Where an interactive render_stroke call spends its wall time: cProfile of 200 calls (host side) - run under
rocprofv3 --kernel-trace --stats for the device side."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
from brushstroke_engine_amd.networks import Generator
res, level = 256, 2
cfg = cfgmod.style1_config(res)
G = Generator(cfg, wmod.random_state_dict(cfg, 0)).to("cuda")
ops = painting.TileOps(G, encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5)))
helper = painting.PaintingHelper(ops)
helper.make_new_canvas(2048, 2048, feature_blending=level)
opts = painting.GanBrushOptions()
opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
rs = np.random.RandomState(0)
def one():
    patch = np.zeros((res, res, 4), np.uint8)
    y0 = rs.randint(10, res - 30)
    patch[y0:y0 + 12, 10:res - 10, 3] = 255
    x, y = int(rs.randint(0, 2048 - res)), int(rs.randint(0, 2048 - res))
    opts.set_position(x, y)
    return helper.render_stroke(patch, None, opts, meta={"x": x, "y": y, "crop_margin": 10})
for _ in range(30): one()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(200): one()
dt = time.perf_counter() - t0
pr.disable()
print(f"mean {dt / 200 * 1e3:.3f} ms per call")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
