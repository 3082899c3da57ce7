import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from brushstroke_engine_amd import _lib, ops
up, ci, co, res, c_next = 2, 128, 64, 64, 64
rs = np.random.RandomState(ci + co + up)
n = 3
hin = res // 2
x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32)).cuda()
w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, c_next)).astype(np.float32)).cuda()
dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
xh, wp = ops.pack_h2(x, st), ops.pack_conv_weight_h3(w)
lib, S = _lib.lib(), torch.cuda.current_stream().cuda_stream
y = torch.empty([n, co, res, res], device="cuda")
_lib.check(lib.nb_modconv3x3_up2_h3(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(), y.data_ptr(), n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "f32")
out = torch.zeros(ops.h2_shape(n, c_next, res, res), dtype=torch.float16, device="cuda")
_lib.check(lib.nb_modconv3x3_up2_h3_h2(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(), nst.data_ptr(), c_next, out.data_ptr(), c_next, n, hin, hin, co, 0.2, 1.4142135, 256.0, S), "h2")
ref = ops.pack_h2(y, nst[:, :co].contiguous())
d = (out != ref)
print("mismatch count", int(d.sum()), "of", d.numel())
idx = d.nonzero()[:20].cpu().numpy()
print(idx)
a = ops.unpack_h2(out, co); b = ops.unpack_h2(ref, co)
print("max abs diff", float((a - b).abs().max()))
for dim, name in enumerate(["n", "cg", "hl", "y", "x", "j"]):
    print(name, np.unique(d.nonzero()[:, dim].cpu().numpy())[:40])
for k in range(0, 12, 2):
    i = d.nonzero()[k].cpu().numpy()
    nn, cg, hl, yy, xx, j = i
    ch = cg * 8 + j
    print("idx", i, "y", y[nn, ch, yy, xx].item().hex(), "nst", nst[nn, ch].item().hex(), "prod", (y[nn, ch, yy, xx] * nst[nn, ch]).item().hex(),
          "out hi/lo", out[nn, cg, 0, yy, xx, j].item(), out[nn, cg, 1, yy, xx, j].item(), "ref hi/lo", ref[nn, cg, 0, yy, xx, j].item(), ref[nn, cg, 1, yy, xx, j].item())
