"""The geometry encoder alone (batch 32, R=256, arithmetic as the painting engine uses it), for kernel traces:
   rocprofv3 --kernel-trace -d gpurun_out/enctrace -o enc --output-format csv -- python3 tools/trace_encoder.py
   python3 tools/trace_encoder_summary.py gpurun_out/enctrace"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import encoder as encmod, synthetic, config as cfgmod
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
dev = torch.device("cuda:0")
R, B = int(os.environ.get("NB_R", "256")), int(os.environ.get("NB_B", "32"))
enc = encmod.HipGeometryEncoder(encmod.random_encoder_state_dict(5), device=dev)
enc.arith = os.environ.get("NB_MODE", "f8")
cfg = cfgmod.style1_config(R)
geom = torch.from_numpy(synthetic.stroke_masks(cfg, B, seed=0)).to(dev)
for _ in range(20): enc.encode(geom)
torch.cuda.synchronize()
N = int(os.environ.get("NB_STEPS", "40"))
t0 = time.perf_counter()
for _ in range(N): enc.encode(geom)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"encoder: {dt / N * 1e3:.3f} ms per batch of {B} at R={R} ({enc.arith})")
