import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brushstroke_engine_amd import ops, _lib
from oracle import neube_oracle as orc
dev = torch.device("cuda:0")
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n, ic, oc, h = 2, 144, 128, 32
rs = np.random.RandomState(ic + oc + h)
x = rs.randn(n, ic, h, h).astype(np.float32); wt = rs.randn(oc, ic, 3, 3).astype(np.float32)
s = (1 + 0.5 * rs.randn(n, ic)).astype(np.float32); s_next = (1 + 0.5 * rs.randn(n, oc)).astype(np.float32)
b = (0.1 * rs.randn(oc)).astype(np.float32); noise = (0.1 * rs.randn(n, 1, 2 * h, 2 * h)).astype(np.float32)
T = torch.from_numpy
want = orc.modulated_conv2d(T(x), T(wt), T(s), noise=T(noise), up=2, padding=1, resample_filter=orc.setup_filter(), flip_weight=False)
want = orc.bias_act(want, T(b), act="lrelu", gain=np.sqrt(2), clamp=256.0) * T(s_next)[:, :, None, None]
wd, sd_ = D(wt), D(s)
wpk, wsq = ops.pack_conv_weight(wd)
d = (sd_.square() @ wsq + 1e-8).rsqrt()
out = torch.zeros(ops.h2_shape(n, oc, 2 * h, 2 * h), dtype=torch.float16, device=dev)
xd, nd, bd, snd = D(x), D(noise), D(b), D(s_next)
rc = _lib.lib().nb_modconv3x3_up2_f32_h2(xd.data_ptr(), ic, None, 0, wpk.data_ptr(), sd_.data_ptr(), d.data_ptr(), nd.data_ptr(), 4 * h * h, bd.data_ptr(), snd.data_ptr(),
                                         out.data_ptr(), n, h, h, oc, 0.2, float(np.sqrt(2)), 256.0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
hi = out[:, :, 0].float().permute(0, 1, 4, 2, 3).reshape(n, -1, 2 * h, 2 * h)[:, :oc].cpu()
lo = out[:, :, 1].float().permute(0, 1, 4, 2, 3).reshape(n, -1, 2 * h, 2 * h)[:, :oc].cpu()
got = hi + lo
e = (got - want).abs()
print("max err", float(e.max()), "err of hi alone", float((hi - want).abs().max()), "lo abs max", float(lo.abs().max()), "frac lo==0", float((lo == 0).float().mean()))
bad = torch.nonzero(e > 5e-5)
print("bad count", bad.shape[0], "of", e.numel(), "channels", sorted(set(bad[:, 1].tolist()))[:20], "rows", sorted(set(bad[:, 2].tolist()))[:10])
resid = want - hi
print("corr(lo, want-hi) on bad:", float((lo[e > 5e-5] * resid[e > 5e-5]).sum() / (resid[e > 5e-5] ** 2).sum()) if bad.shape[0] else None)
