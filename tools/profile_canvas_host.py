"""Host-side profile of one steady-state PaintingHelper.paint_image call on the 4096^2 synthetic drawing (BASELINE config 3):
cProfile of the Python that drives the device (the device work is asynchronous; what shows here is issue time + waits).
    gpurun -- 'python tools/profile_canvas_host.py'"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_canvas
from brushstroke_engine_amd import config as cfgmod, encoder as encmod, painting, weights as wmod
from brushstroke_engine_amd.networks import Generator
dev = torch.device("cuda:0")
geom = bench_canvas.synthetic_drawing(4096, 4096)
cfg = cfgmod.style1_config(256)
G = Generator(cfg, wmod.random_state_dict(cfg, seed=0), conv_mode="f8").to(dev)
enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5), device=dev)
ops = painting.TileOps(G, enc)
helper = painting.PaintingHelper(ops, batch=32)
helper.set_feature_blending(2)
opts = painting.GanBrushOptions()
opts.set_style(torch.from_numpy(np.random.RandomState(594).randn(1, cfg.z_dim)), 594)
for _ in range(3):
    helper.paint_image(geom, opts, crop_margin=10)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); helper.paint_image(geom, opts, crop_margin=10); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("paint_image returns after %.1f ms; device idle after %.1f ms" % (1e3 * np.mean([a for a, _ in ts]), 1e3 * np.mean([b for _, b in ts])))
pr = cProfile.Profile()
pr.enable(); helper.paint_image(geom, opts, crop_margin=10); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
