"""Developer tool: time the upfirdn2d configurations of the training step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import ops
dev = torch.device("cuda")
f = ops.setup_filter((1, 3, 3, 1), device=dev)
for shape, up, down, pad, ff in (((8, 64, 256, 256), 1, 1, [2, 2, 2, 2], f), ((8, 64, 257, 257), 1, 1, [1, 1, 1, 1], f), ((8, 128, 64, 64), 2, 1, [2, 1, 2, 1], f),
                                 ((8, 64, 256, 256), 1, 2, [1, 1, 1, 1], f), ((8, 128, 128, 128), 2, 1, [0, -1, 0, -1], None)):
    x = torch.randn(*shape, device=dev)
    for _ in range(3): y = ops.upfirdn2d(x, ff, up=up, down=down, padding=pad)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = ops.upfirdn2d(x, ff, up=up, down=down, padding=pad)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f"{shape} up {up} down {down}: {dt * 1e6:.0f} us, {gb / dt / 1e3:.2f} TB/s")
