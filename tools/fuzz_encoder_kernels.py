"""Randomised shape sweep: the small-tile encoder conv kernel against the large-tile one (same C entry, debug hook picks
the kernel): stride 1 / 2, fp32 / H2 output, widths the large kernel tiles (16 or multiples of 32), ragged c_out."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, encoder as encmod
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(int(os.environ.get("NB_SEED", "0")))

def to_h2(x):
    n, c, h, w = x.shape
    hi = x.half(); lo = (x - hi.float()).half()
    return torch.stack([hi, lo], 1).reshape(n, 2, c // 8, 8, h, w).permute(0, 2, 1, 4, 5, 3).contiguous()

worst, cases = 0.0, 0
for it in range(int(os.environ.get("NB_CASES", "200"))):
    stride = int(rs.choice([1, 2]))
    wo = int(rs.choice([16, 32, 64]))
    ho = int(rs.choice([16, 32, 64])) if wo != 16 else int(rs.choice([16, 32]))
    h, w = ho * stride, wo * stride
    n = int(rs.randint(1, 4))
    ci = 16 * int(rs.randint(1, 17))
    h2out = bool(rs.randint(0, 2))
    co = int(rs.choice([16, 32, 64, 128, 256])) if h2out else int(rs.choice([8, 16, 40, 128, 200, 256]))
    x = torch.from_numpy(rs.randn(n, ci, h, w).astype(np.float32)).cuda()
    wt = (rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)
    b = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    xd = to_h2(x); wd = torch.from_numpy(encmod.pack_enc_weight_h3(wt)).cuda()
    outs = []
    for small in (0, 1):
        lib.nb_debug_set_enc_small(small)
        if h2out:
            y = torch.zeros([n, co // 8, 2, ho, wo, 8], dtype=torch.float16, device="cuda")
            _lib.check(lib.nb_enc_conv3x3_h3(xd.data_ptr(), ci, wd.data_ptr(), b.data_ptr(), None, y.data_ptr(), n, h, w, co, stride, 0.01, S), "enc")
            outs.append(y.float()[:, :, 0] + y.float()[:, :, 1])
        else:
            y = torch.full([n, co, ho, wo], float("nan"), device="cuda")
            _lib.check(lib.nb_enc_conv3x3_h3(xd.data_ptr(), ci, wd.data_ptr(), b.data_ptr(), y.data_ptr(), None, n, h, w, co, stride, 0.01, S), "enc")
            outs.append(y)
    lib.nb_debug_set_enc_small(-1)
    torch.cuda.synchronize()
    err = float((outs[0] - outs[1]).abs().max()); scale = max(1.0, float(outs[0].abs().max()))
    assert err == err and err <= 2e-5 * scale, (it, stride, n, ci, co, h, w, h2out, err)
    worst = max(worst, err / scale); cases += 1
print(f"{cases} cases ok, worst relative difference {worst:.2e}")
