"""Randomised shape sweep of the small-image split-f16 kernels against the fp32-MFMA kernels (same C ABI, same inputs):
up = 1 and up = 2 (with and without a concatenated second input), odd batches, ragged c_out, shared / per-sample / no noise."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(int(os.environ.get("NB_SEED", "0")))
worst, cases = 0.0, 0
for it in range(int(os.environ.get("NB_CASES", "300"))):
    up = int(rs.choice([1, 2]))
    h = int(rs.choice([4, 8, 16, 32, 64] if up == 1 else [4, 8, 16, 32]))
    n = int(rs.randint(1, 6))
    c1 = 16 * int(rs.randint(1, 13))
    c2 = 16 * int(rs.randint(0, 3)) if up == 2 else 0
    co = int(rs.choice([8, 24, 32, 40, 64, 96, 128, 136, 200]))
    ci = c1 + c2
    x = torch.from_numpy(rs.randn(n, ci, h, h).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    ho = h * up
    nmode = int(rs.randint(0, 3))            # 0 none, 1 shared, 2 per sample
    noise = None if nmode == 0 else torch.from_numpy(rs.randn(1 if nmode == 1 else n, ho, ho).astype(np.float32)).cuda()
    nstride = ho * ho if nmode == 2 else 0
    clamp = float(rs.choice([-1.0, 1.5, 256.0]))
    x1 = x[:, :c1].contiguous(); x2 = x[:, c1:].contiguous() if c2 else None
    wpk, _ = ops.pack_conv_weight(w)
    ref = torch.empty(n, co, ho, ho, device="cuda"); got = torch.full_like(ref, float("nan"))
    P = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.nb_modconv3x3_f32(P(x1), c1, P(x2), c2, P(wpk), P(st), P(dco), P(noise), nstride, P(bias), P(ref), n, h, h, co, up,
                                     0.2, 1.4142135, clamp, S), "f32")
    if up == 1:
        w3 = ops.pack_conv_weight_h3(w)
        _lib.check(lib.nb_modconv3x3_up1_small_h3(P(x1), c1, P(w3), P(st), P(dco), P(noise), nstride, P(bias), P(got), n, h, h, co,
                                                  0.2, 1.4142135, clamp, S), "small up1")
    else:
        f = ops.setup_filter([1, 3, 3, 1], device="cuda")
        w3 = ops.pack_conv_weight_h3_up2_phases(w, f)
        _lib.check(lib.nb_modconv3x3_up2_small_h3(P(x1), c1, P(x2), c2, P(w3), P(st), P(dco), P(noise), nstride, P(bias), P(got), n, h, h, co,
                                                  0.2, 1.4142135, clamp, S), "small up2")
    torch.cuda.synchronize()
    err = float((got - ref).abs().max()); scale = max(1.0, float(ref.abs().max()))
    assert err == err and err <= 3e-5 * scale, (it, up, n, c1, c2, co, h, nmode, clamp, err, scale)
    worst = max(worst, err / scale); cases += 1
print(f"{cases} cases ok, worst relative error {worst:.2e}")
