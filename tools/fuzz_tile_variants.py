"""Randomised shape sweep: the two tile forms of the large split-f16 kernels (up=2: 12 / 5 quad rows, up=1: 2 / 1 pixel
rows per wave) must give bit-identical fp32 and hand-off outputs in both operand formats."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(int(os.environ.get("NB_SEED", "0")))
cases = 0
for it in range(int(os.environ.get("NB_CASES", "120"))):
    up = int(rs.choice([1, 2]))
    fmt = int(rs.randint(0, 2))
    n = int(rs.randint(1, 4))
    ci = 16 * int(rs.randint(1, 13))
    co = 16 * int(rs.randint(1, 9))
    if up == 1:
        h = 16 * int(rs.randint(1, 5)); w = 32 * int(rs.randint(1, 3))
    else:
        h = int(rs.choice([8, 12, 16, 24, 32, 40])); w = 32 * int(rs.randint(1, 3))
    ho, wo = h * up, w * up
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, ho, wo).astype(np.float32)).cuda()
    xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
    wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(wt)
    res = []
    for variant in (0, 1):
        if up == 2: lib.nb_debug_set_up2_tile(12 if variant == 0 else 5)
        else: lib.nb_debug_set_up1_rows(2 if variant == 0 else 1)
        y = torch.empty([n, co, ho, wo], device="cuda")
        out = torch.zeros(ops.h2_shape(n, co, ho, wo), dtype=torch.float16, device="cuda")
        common = (dco.data_ptr(), noise.data_ptr(), ho * wo, bias.data_ptr())
        if up == 2:
            _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, fmt, 0, n, h, w, co, 0.2, 1.4142135, 256.0, S), "a")
            _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), co, co, fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S), "b")
        else:
            _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, y.data_ptr(), None, None, 0, 0, None, fmt, 0, n, h, w, co, 0.2, 1.4142135, 256.0, S), "a")
            _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), *common, None, out.data_ptr(), nst.data_ptr(), co, co, None, fmt, fmt, n, h, w, co, 0.2, 1.4142135, 256.0, S), "b")
        torch.cuda.synchronize()
        res.append((y, out))
    lib.nb_debug_set_up2_tile(0); lib.nb_debug_set_up1_rows(0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (it, up, fmt, n, ci, co, h, w)
    assert not torch.isnan(res[0][0]).any()
    cases += 1
print(f"{cases} cases ok: tile forms bit-identical")
