"""Race hunt for the ping-pong K loop of the up=1 f8 kernel: the two large up=1 launches of the BASELINE step, repeated, each result compared
bit for bit with the software-pipelined loop's (a ring hazard would show as a sporadic difference).   gpurun -- 'python tools/stress_up1_pp.py'"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)
lib = _lib.lib()
reps = int(os.environ.get("NB_REPS", "300"))
S = torch.cuda.current_stream().cuda_stream
bad = 0
for ci, co, res, n in ((128, 128, 128, 32), (64, 64, 256, 32), (48, 64, 64, 8), (16, 128, 32, 4)):
    rs = np.random.RandomState(ci)
    x = torch.from_numpy(rs.randn(n, ci, res, res).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
    noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
    xh, wp = ops.pack_h2f8(x, st), ops.pack_conv_weight_h3f8(w)
    del x
    outs = [torch.zeros(ops.h2_shape(n, co, res, res), dtype=torch.int16, device="cuda") for _ in range(2)]

    def run(pp, out):
        lib.nb_debug_set_up1_pp(pp)
        _lib.check(lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(), None, out.data_ptr(),
                                               nst.data_ptr(), co, co, None, 1, 1, n, res, res, co, 0.2, 1.4142135, 256.0, S), "up1")
    run(0, outs[0]); torch.cuda.synchronize()
    ref = outs[0].clone()
    diffs = 0
    for r in range(reps):
        outs[1].zero_()
        run(1, outs[1])
        if r % 3 == 0:                       # (some repetitions with another kernel right behind: different arrival patterns)
            run(0, outs[0])
        if not torch.equal(outs[1], ref):
            diffs += 1
    lib.nb_debug_set_up1_pp(-1)
    print(f"up1 {ci}->{co}@{res} n={n}: {reps} repetitions of the ping-pong loop, {diffs} differ from the software-pipelined loop's result", flush=True)
    bad += diffs
    del xh, outs, ref
    torch.cuda.empty_cache()
print("OK" if bad == 0 else f"FAILED: {bad} differing results")
sys.exit(1 if bad else 0)
