"""Run-to-run determinism of the f8 up=1 kernel variants (same inputs, 5 launches each, fp32 and f8-H2 outputs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
for (n, ci, co, res) in [(16, 64, 64, 256), (16, 128, 128, 128), (16, 128, 128, 64), (16, 128, 128, 32), (1, 64, 64, 256), (1, 128, 128, 128)]:
    rs = np.random.RandomState(ci + co + res)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32) * 2).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.ones(n, co, device="cuda"); bias = torch.zeros(co, device="cuda")
    xh, wp = ops.pack_h2f8(x, st), ops.pack_conv_weight_h3f8(w)
    outs = []
    for rep in range(5):
        y = torch.zeros([n, co, res, res], device="cuda")
        _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), y.data_ptr(), None, None, 0, 0,
                                               None, 1, 0, n, res, res, co, 0.2, 1.4142135, 256.0, S), "f32")
        torch.cuda.synchronize()
        outs.append(y)
    diffs = [int((outs[0] != o).sum()) for o in outs[1:]]
    ref = torch.nn.functional.conv2d((x * st[:, :, None, None])[:2].double().cpu(), w.double().cpu(), padding=1)
    act = torch.nn.functional.leaky_relu(ref, 0.2) * 1.4142135
    print(f"n={n} {ci}->{co}@{res}: mismatching elements vs first launch {diffs}; max err vs float64 {float((outs[0][:2].cpu().double() - act.clamp(-256, 256)).abs().max()):.2e}")
