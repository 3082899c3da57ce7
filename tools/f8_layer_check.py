"""Stage check of the f8 operand format: one conv1 layer against float64, and its time against the h3 kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
for (n, ci, co, res) in [(2, 64, 64, 64), (32, 64, 64, 256), (32, 128, 128, 128)]:
    rs = np.random.RandomState(ci + co)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32) * 2).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    dco = torch.ones(n, co, device="cuda"); bias = torch.zeros(co, device="cuda")
    outs = {}
    for mode in ("h3", "f8"):
        xh = (ops.pack_h2 if mode == "h3" else ops.pack_h2f8)(x, st)
        wp = (ops.pack_conv_weight_h3 if mode == "h3" else ops.pack_conv_weight_h3f8)(w)
        fn = lib.nb_modconv3x3_up1_h3 if mode == "h3" else lib.nb_modconv3x3_up1_h3f8
        y = torch.empty([n, co, res, res], device="cuda")
        def launch():
            _lib.check(fn(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), y.data_ptr(), n, res, res, co,
                          1.0, 1.0, -1.0, S), mode)
        for _ in range(3): launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): launch()
        e1.record(); torch.cuda.synchronize()
        outs[mode] = (y.clone(), e0.elapsed_time(e1) / 10)
    if n <= 2:
        ref = torch.nn.functional.conv2d((x * st[:, :, None, None]).double().cpu(), w.double().cpu(), padding=1)
        for mode in outs:
            print(f"  {mode}: max abs err vs float64 {float((outs[mode][0].cpu().double() - ref).abs().max()):.3e} (|y| max {float(ref.abs().max()):.2f})")
    print(f"n={n} {ci}->{co}@{res}: h3 {outs['h3'][1]:.4f} ms  f8 {outs['f8'][1]:.4f} ms  ({outs['h3'][1] / outs['f8'][1]:.2f}x),"
          f" f8 vs h3 max diff {float((outs['f8'][0] - outs['h3'][0]).abs().max()):.3e}")
