"""Run-to-run determinism of the up=2 split-f16 kernel (H2 and f8 operands, hand-off output), launched on two streams at
once so that workgroups of different launches share the chip: mismatching output elements vs the first launch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for fmt in (0, 1):
    for (n, ci, co, hin) in [(16, 128, 64, 128), (16, 384, 128, 64), (16, 128, 128, 32), (16, 128, 128, 16), (1, 128, 64, 128)]:
        rs = np.random.RandomState(ci + co + hin)
        res = 2 * hin
        x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32) * 2).cuda()
        w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
        st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
        nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
        dco = torch.ones(n, co, device="cuda"); bias = torch.zeros(co, device="cuda")
        noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
        xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
        wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(w)
        outs = [[], []]
        torch.cuda.synchronize()
        for rep in range(6):
            for si, s in enumerate(streams):
                with torch.cuda.stream(s):
                    out = torch.zeros(ops.h2_shape(n, co, res, res), dtype=torch.float16, device="cuda")
                    _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), noise.data_ptr(), res * res, bias.data_ptr(),
                                                           None, out.data_ptr(), nst.data_ptr(), co, co, fmt, fmt, n, hin, hin, co, 0.2, 1.4142135, 256.0,
                                                           s.cuda_stream), "up2")
                    outs[si].append(out)
        torch.cuda.synchronize()
        ref = outs[0][0].view(torch.int16)
        diffs = [int((ref != o.view(torch.int16)).sum()) for so in outs for o in so]
        print(f"fmt {fmt} n={n} {ci}->{co}@{res}: mismatching elements vs first launch {diffs[1:]}")
        if any(diffs) and os.environ.get("NB_DET_VERBOSE"):
            for so in outs:
                for o in so:
                    bad = (ref != o.view(torch.int16)).nonzero()
                    if len(bad):
                        # H2 layout [n][c8][2][H][W][8]
                        print("   first mismatches (n, c8, hi/lo, y, x, ch):", bad[:6].tolist(), "... rows", sorted(set(bad[:, 3].tolist()))[:12], "cols", sorted(set(bad[:, 4].tolist()))[:12],
                              "planes", sorted(set(bad[:, 2].tolist())), "c8", sorted(set(bad[:, 1].tolist())), "ch", sorted(set(bad[:, 5].tolist())))
