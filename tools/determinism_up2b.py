import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
import ctypes
lib = _lib.lib()
lib.nb_debug_set_up2_tile.argtypes = [ctypes.c_int]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for tile in [int(t) for t in os.environ.get("NB_TILES", "0 5 12").split()]:
  lib.nb_debug_set_up2_tile(tile)
  for fmt, ofmt in ((0, 0), (0, 1), (1, 0), (1, 1)):
    for (n, ci, co, hin) in [(1, 128, 64, 128), (4, 128, 64, 128)]:
        rs = np.random.RandomState(ci + co + hin)
        res = 2 * hin
        x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32) * 2).cuda()
        w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
        st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
        nst = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda()
        dco = torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda(); bias = torch.from_numpy(rs.randn(co).astype(np.float32)).cuda()
        noise = torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda()
        xh = (ops.pack_h2f8 if fmt else ops.pack_h2)(x, st)
        wp = (ops.pack_conv_weight_h3f8 if fmt else ops.pack_conv_weight_h3)(w)
        outs = [[], []]
        torch.cuda.synchronize()
        for rep in range(8):
            for si, s in enumerate(streams):
                with torch.cuda.stream(s):
                    out = torch.zeros(ops.h2_shape(n, co, res, res), dtype=torch.float16, device="cuda")
                    _lib.check(lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), (None if os.environ.get("NB_NO_NOISE") else noise.data_ptr()), res * res, bias.data_ptr(),
                                                           None, out.data_ptr(), nst.data_ptr(), co, co, fmt, ofmt, n, hin, hin, co, 0.2, 1.4142135, 256.0,
                                                           s.cuda_stream), "up2")
                    outs[si].append(out)
        torch.cuda.synchronize()
        ref = outs[0][0].view(torch.int16)
        diffs = [int((ref != o.view(torch.int16)).sum()) for so in outs for o in so]
        print(f"tile {tile} in_fmt {fmt} out_fmt {ofmt} n={n} {ci}->{co}@{res}: {diffs[1:]}")
