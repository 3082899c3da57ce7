"""Per-workgroup phase timeline of the geometry encoder's large-tile conv kernel (debug hook nb_debug_set_enc_timestamps): the three
stride-2 layers of a batch of 32 at R=256 with f8 operands, called directly through the C ABI on random operands (timing only).
    gpurun -- 'python tools/phase_times_enc.py'"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.nb_debug_set_enc_timestamps.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.nb_debug_set_enc_timestamps.restype = None
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = int(os.environ.get("NB_B", "32"))
cap = 8192
for (ci, co, r, stride) in [(64, 128, 256, 2), (128, 256, 128, 2), (256, 256, 64, 2), (16, 256, 32, 1)]:
    ro = r // stride
    x = (torch.randn(n * ci * 2 * r * r, device=dev) * 0.1).half()
    nch, co_ld = (ci + 15) // 16, (co + 127) // 128 * 128
    w = (torch.randn(nch * 3 * 3 * 2 * 2 * co_ld * 8, device=dev) * 0.05).half()
    b = torch.zeros(co, device=dev)
    y = torch.empty(n * co * 2 * ro * ro, device=dev, dtype=torch.float16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def launch():
        rc = lib.nb_enc_conv3x3_ex(P(x), ci, P(w), P(b), None, P(y), None, 0, co // 8, 0, 1, 1, n, r, r, co, stride, ctypes.c_float(0.01), st)
        assert rc == 0, lib.nb_last_error()
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1)
    ts = torch.zeros(cap, 8, dtype=torch.int64, device=dev)
    lib.nb_debug_set_enc_timestamps(P(ts), cap)
    launch(); torch.cuda.synchronize()
    lib.nb_debug_set_enc_timestamps(None, 0)
    t = ts.cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    us = (t - t[:, 0].min()) / 100.0
    have1 = (t[:, 1] > 0).all()
    print(f"enc conv {ci}->{co} in {r}^2 stride {stride} n={n}: {t.shape[0]} workgroups, kernel {plain * 1e3:.1f} us, span {us[:, 4].max():.1f} us, "
          f"workgroup mean {(us[:, 4] - us[:, 0]).mean():.1f} us; steps {nch * 3}")
    if have1:
        pro, kl = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1]
        print(f"    prologue {pro.mean():6.2f} us   k-loop {kl.mean():6.2f} us = {kl.mean() / (nch * 3):.3f} us per step")
    else:
        print(f"    prologue + k-loop {(us[:, 2] - us[:, 0]).mean():6.2f} us")
    print(f"    epilogue to LDS {(us[:, 3] - us[:, 2]).mean():6.2f} us   stores {(us[:, 4] - us[:, 3]).mean():6.2f} us")

# the fused stem + first stride-2 stage (nb_enc_stem_conv3x3_f8): slot 5 = strips built, slot 1 = chunk 0's slabs computed (the other chunks'
# slabs ride in the gaps of the steps)
r, co = 256, 128
img = torch.rand(n, 1, r, r, device=dev)
w50 = torch.randn(64, 50, device=dev) * 0.1
b0 = torch.zeros(64, device=dev)
w = (torch.randn(4 * 3 * 3 * 2 * 2 * 128 * 8, device=dev) * 0.05).half()
b = torch.zeros(co, device=dev)
y = torch.empty(n * co * 2 * (r // 2) ** 2, device=dev, dtype=torch.float16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch():
    rc = lib.nb_enc_stem_conv3x3_f8(P(img), P(w50), P(b0), 0, P(w), P(b), P(y), 1, n, r, r, co, ctypes.c_float(0.01), st)
    assert rc == 0, lib.nb_last_error()
for _ in range(3): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); e1.record(); torch.cuda.synchronize()
plain = e0.elapsed_time(e1)
ts = torch.zeros(cap, 8, dtype=torch.int64, device=dev)
lib.nb_debug_set_enc_timestamps(P(ts), cap)
launch(); torch.cuda.synchronize()
lib.nb_debug_set_enc_timestamps(None, 0)
t = ts.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
us = (t - t[:, 0].min()) / 100.0
print(f"fused stem + conv 64->{co} in {r}^2 stride 2 n={n}: {t.shape[0]} workgroups, kernel {plain * 1e3:.1f} us, span {us[:, 4].max():.1f} us, "
      f"workgroup mean {(us[:, 4] - us[:, 0]).mean():.1f} us; steps 12")
print(f"    strips + stem weights {(us[:, 5] - us[:, 0]).mean():6.2f} us   chunk 0's slabs {(us[:, 1] - us[:, 5]).mean():6.2f} us   "
      f"k-loop {(us[:, 2] - us[:, 1]).mean():6.2f} us = {(us[:, 2] - us[:, 1]).mean() / 12:.3f} us per step (with the stem of the next chunk in its gaps)")
print(f"    epilogue {(us[:, 3] - us[:, 2]).mean():6.2f} us")
