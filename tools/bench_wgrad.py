import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
for split in (False, True):
    ops.WGRAD_SPLIT_F16 = split
    for (n, c, h) in ((8, 128, 128), (8, 64, 256), (8, 128, 64)):
        u = torch.randn(n, c, h, h, device="cuda"); v = torch.randn(n, c, h, h, device="cuda")
        for _ in range(3): ops.conv2d_wgrad(u, v, 1, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.conv2d_wgrad(u, v, 1, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        fl = 2.0 * n * c * c * 9 * h * h
        print(f"split_f16={split} n={n} c={c} h={h}: {dt*1e6:.0f} us  {fl/dt/1e12:.1f} TFLOP/s")
