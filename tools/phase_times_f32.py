"""Per-workgroup phase timeline of the fp32 split-K up=1 kernel at batch 1 (debug hook nb_debug_set_timestamps_f32)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
lib = _lib.lib()
lib.nb_debug_set_timestamps_f32.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.nb_debug_set_timestamps_f32.restype = None
for res in (4, 8, 16, 32, 64):
    n, ci, co = 1, 128, 128
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.ones(n, ci, device="cuda"); dco = torch.ones(n, co, device="cuda"); bias = torch.zeros(co, device="cuda")
    noise = torch.zeros(n, res, res, device="cuda")
    wpk, wsq = ops.pack_conv_weight(w)
    y = torch.empty(n, co, res, res, device="cuda")
    S = torch.cuda.current_stream().cuda_stream
    def launch():
        _lib.check(lib.nb_modconv3x3_f32(x.data_ptr(), ci, None, 0, wpk.data_ptr(), st.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res,
                                         bias.data_ptr(), y.data_ptr(), n, res, res, co, 1, 0.2, 1.4142135, 256.0, S), "conv")
    for _ in range(5): launch()
    torch.cuda.synchronize()
    # cold-ish timing like inside the step: evict by touching a big buffer first
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    ts = torch.zeros([4096, 8], dtype=torch.int64, device="cuda")
    lib.nb_debug_set_timestamps_f32(ts.data_ptr(), 4096)
    big.fill_(1.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    lib.nb_debug_set_timestamps_f32(None, 0)
    t = ts.cpu().numpy().astype(np.float64); t = t[t[:, 0] > 0]
    d = np.diff(t[:, :6], axis=1) / 100.0          # us
    print(f"res {res:3d}: {len(t):4d} WGs, launch (events) {e0.elapsed_time(e1) * 1e3:6.1f} us | mean per WG: prologue {d[:,0].mean():5.2f}  K loop {d[:,1].mean():5.2f}  "
          f"reduce {d[:,2].mean():5.2f}  epilogue {d[:,3].mean():5.2f}  drain {d[:,4].mean():5.2f} | first start -> last end {(t[:,5].max() - t[:,0].min()) / 100:6.2f} us")
