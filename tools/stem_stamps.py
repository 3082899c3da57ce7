import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from brushstroke_engine_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib()
lib.nb_debug_set_enc_timestamps.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.nb_debug_set_enc_timestamps.restype = None
P = lambda t: ctypes.c_void_p(t.data_ptr())
n, r, co, cap = 32, 256, 128, 8192
img = torch.rand(n, 1, r, r, device=dev); w50 = torch.randn(64, 50, device=dev) * 0.1; b0 = torch.zeros(64, device=dev)
w = (torch.randn(4 * 3 * 3 * 2 * 2 * 128 * 8, device=dev) * 0.05).half(); b = torch.zeros(co, device=dev)
y = torch.empty(n * co * 2 * (r // 2) ** 2, device=dev, dtype=torch.float16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch():
    assert lib.nb_enc_stem_conv3x3_f8(P(img), P(w50), P(b0), 0, P(w), P(b), P(y), 1, n, r, r, co, ctypes.c_float(0.01), st) == 0
for _ in range(3): launch()
ts = torch.zeros(cap, 8, dtype=torch.int64, device=dev)
lib.nb_debug_set_enc_timestamps(P(ts), cap); launch(); torch.cuda.synchronize(); lib.nb_debug_set_enc_timestamps(None, 0)
t = ts.cpu().numpy().astype(np.float64); t = t[t[:, 0] > 0]
print("O_0 slab back to back, wave 0, s_memtime ticks (100 MHz?) per type: L %.1f  M %.1f  S %.1f  (mean over workgroups); 6 L, 5 M... ops" % tuple(t[:, 5:8].mean(0)))
print("per op: L %.1f M %.1f S %.1f" % (t[:,5].mean()/5, t[:,6].mean()/5, t[:,7].mean()/5))
