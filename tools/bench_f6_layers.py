"""Same-box A/B of the big launches with f8 and f6 operands (hand-off output in the f8 format for both, so the epilogue is the same and the
difference is the K loop):  gpurun -- 'python tools/bench_f6_layers.py'"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)

lib = _lib.lib()
n = int(os.environ.get("NB_N", "32"))
LAYERS = [(1, 128, 128, 128, 128), (1, 64, 64, 256, 64), (2, 384, 128, 128, 128), (2, 128, 64, 256, 64)]      # up, ci, co, out res, c_next
if os.environ.get("NB_LAYERS"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["NB_LAYERS"].split(",")]
fmts = [int(f) for f in os.environ.get("NB_FMTS", "1,2").split(",")]
rs = np.random.RandomState(0)
S = torch.cuda.current_stream().cuda_stream
for up, ci, co, res, c_next in LAYERS:
    hin = res if up == 1 else res // 2
    x = torch.from_numpy((rs.randn(n, ci, hin, hin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, c_next)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
    out = torch.zeros(ops.h2_shape(n, c_next, res, res), dtype=torch.float16, device="cuda")
    ops_ = {}
    for fmt in fmts:
        try:
            xh = (ops.pack_h2f8 if fmt == 1 else ops.pack_h2f6)(x, st)
            wp = (ops.pack_conv_weight_h3f8 if fmt == 1 else ops.pack_conv_weight_h3f6)(w)
            ops_[fmt] = (xh, wp)
        except Exception as e:
            print("pack failed", fmt, e)
    del x

    def run(fmt, out_fmt=1):
        xh, wp = ops_[fmt]
        common = (dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr())
        if up == 1:
            return lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), c_next, c_next, None,
                                               fmt, out_fmt, n, hin, hin, co, 0.2, 1.4142135, 256.0, S)
        return lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), c_next, c_next,
                                           fmt, out_fmt, n, hin, hin, co, 0.2, 1.4142135, 256.0, S)
    res_ms = {f: [] for f in ops_}
    ok = {}
    for fmt in list(ops_):
        rc = run(fmt)
        ok[fmt] = rc == 0
        if rc != 0:
            print(f"up{up} {ci}->{co}@{res} fmt {fmt}: not supported ({_lib.last_error() if hasattr(_lib, 'last_error') else rc})")
    for rep in range(5):
        for fmt in ops_:
            if not ok[fmt]:
                continue
            for _ in range(5):
                run(fmt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                run(fmt)
            e1.record(); torch.cuda.synchronize()
            res_ms[fmt].append(e0.elapsed_time(e1) / 40)
    line = f"up{up} {ci}->{co}@{res} n={n}: " + "  ".join(f"fmt{f} {min(v) * 1e3:.1f} us (median {np.median(v) * 1e3:.1f})" for f, v in res_ms.items() if v)
    if all(res_ms.get(f) for f in (1, 2)):
        line += f"   f6/f8 = {min(res_ms[2]) / min(res_ms[1]):.3f}"
    print(line, flush=True)
    del ops_, out
    torch.cuda.empty_cache()
