"""Ping-pong K loop of the up=1 f8 kernel: cycles wave 0 spends in its load segments, compute segments and at the two barriers
(variant build -DNB_PP_STAMPS=1 -DNB_UP1_PP_DEFAULT=1; tools/build_variant.sh ppst nb_modconv_h3.hip "...")."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
lib = _lib.lib()
lib.nb_debug_set_timestamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for ci, co, res in ((128, 128, 128), (64, 64, 256)):
    n = 32
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st, dco, bias = torch.ones(n, ci, device="cuda"), torch.ones(n, co, device="cuda"), torch.zeros(co, device="cuda")
    xh, wp = ops.pack_h2f8(x, st), ops.pack_conv_weight_h3f8(w)
    del x
    out = torch.empty(ops.h2_shape(n, co, res, res), dtype=torch.float16, device="cuda")
    nst = torch.ones(n, co, device="cuda")
    S = torch.cuda.current_stream().cuda_stream
    cap = 1 << 16
    ts = torch.zeros([cap, 8], dtype=torch.int64, device="cuda")
    def launch():
        _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), None, out.data_ptr(), nst.data_ptr(), co, co,
                                               None, 1, 1, n, res, res, co, 0.2, 1.4142135, 256.0, S), "up1")
    for _ in range(3): launch()
    torch.cuda.synchronize()
    lib.nb_debug_set_timestamps(ts.data_ptr(), cap)
    launch(); torch.cuda.synchronize()
    lib.nb_debug_set_timestamps(None, 0)
    t = ts.cpu().numpy()
    t = t[t[:, 0] > 0]
    a, b = t[:, 6].astype(np.uint64), t[:, 7].astype(np.uint64)
    L, C = (a & 0xffffffff).astype(np.float64), (a >> 32).astype(np.float64)
    W1, W2 = (b & 0xffffffff).astype(np.float64), (b >> 32).astype(np.float64)
    steps = (ci // 16) * 3
    kl = (t[:, 2] - t[:, 1]) / 100.0
    print(f"up1 {ci}->{co}@{res}: {len(t)} workgroups, {steps} steps; K loop {kl.mean():.2f} us; wave 0 cycles per step: load {L.mean() / steps:.0f}  wait {W1.mean() / steps:.0f}  "
          f"compute {C.mean() / steps:.0f}  wait {W2.mean() / steps:.0f}  total {(L + C + W1 + W2).mean() / steps:.0f}")
    del ts, out, xh
    torch.cuda.empty_cache()
