import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
for n in (32, 1):
  for res in (4, 8, 16, 32):
    ci = co = 128
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / 34).astype(np.float32)).cuda()
    st = torch.ones(n, ci, device="cuda"); d = torch.ones(n, co, device="cuda"); b = torch.zeros(co, device="cuda")
    wp = ops.pack_conv_weight_h3(w)
    y = torch.empty(n, co, res, res, device="cuda")
    def run():
        _lib.check(lib.nb_modconv3x3_up1_small_h3(x.data_ptr(), ci, wp.data_ptr(), st.data_ptr(), d.data_ptr(), None, 0, b.data_ptr(), y.data_ptr(),
                                                  n, res, res, co, 0.2, 1.414, 256.0, S), "small")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} {res}x{res}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
