"""Per-workgroup phase timeline of the split-f16 conv kernels (debug hook nb_debug_set_timestamps): where a
workgroup's time goes -- prologue DMA, K loop, epilogue through LDS, store issue, store drain.

    python tools/phase_times.py            # on the GPU box
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nb_debug_env; nb_debug_env.apply()          # developer NB_* switches -> the library's debug setters (it reads no environment itself)


FMT = int(os.environ.get("NB_PHASE_FMT", "1" if os.environ.get("NB_PHASE_F8") == "1" else "0"))      # operand format: 0 H2, 1 f8, 2 f6
F8 = FMT != 0
H2OUT = os.environ.get("NB_PHASE_H2OUT") == "1"          # write the consumer's H2 / f8 tensor instead of fp32 NCHW


def run(kind, n, ci, co, res):
    lib = _lib.lib()
    lib.nb_debug_set_timestamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.nb_debug_set_timestamps.restype = None
    rs = np.random.RandomState(0)
    hin = res if kind == "up1" else res // 2
    x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    styles = torch.ones(n, ci, device="cuda")
    dco = torch.ones(n, co, device="cuda")
    bias = torch.zeros(co, device="cuda")
    xh = (ops.pack_h2f6 if FMT == 2 else ops.pack_h2f8 if F8 else ops.pack_h2)(x, styles)
    wp = (ops.pack_conv_weight_h3f6 if FMT == 2 else ops.pack_conv_weight_h3f8 if F8 else ops.pack_conv_weight_h3)(w)
    cap = 1 << 16
    ts = torch.zeros([cap, 8], dtype=torch.int64, device="cuda")
    y = torch.empty([n, co, res, res], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    yh2 = torch.empty(ops.h2_shape(n, co, res, res), dtype=torch.float16, device="cuda") if H2OUT else None
    nst = torch.ones(n, co, device="cuda")
    yp, hp, sp, cn = (None, yh2.data_ptr(), nst.data_ptr(), co) if H2OUT else (y.data_ptr(), None, None, 0)

    def launch():
        if kind == "up1":
            rc = lib.nb_modconv3x3_up1_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), yp,
                                             hp, sp, cn, cn, None, FMT, int(F8 and H2OUT), n, hin, hin, co, 0.2, 1.4142135, 256.0, st)
        else:
            rc = lib.nb_modconv3x3_up2_h3_ex(xh.data_ptr(), ci, wp.data_ptr(), dco.data_ptr(), None, 0, bias.data_ptr(), yp,
                                             hp, sp, cn, cn, FMT, int(F8 and H2OUT), n, hin, hin, co, 0.2, 1.4142135, 256.0, st)
        _lib.check(rc, kind)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1)
    lib.nb_debug_set_timestamps(ts.data_ptr(), cap)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    lib.nb_debug_set_timestamps(None, 0)
    t = ts.cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                                    # 100 MHz ticks -> us
    if kind == "up2":
        # the up=2 kernel has no stamp 3 (its epilogue runs in four rounds; slot 3 carries the packed in-epilogue cycle
        # counts instead): epilogue = stamp 2 -> stamp 4
        us = us[:, [0, 1, 2, 4, 5]]
        names = ["prologue", "k-loop", "epilogue (4 rounds)", "store drain"]
    elif H2OUT:
        if (t[:, 5] > 0).all():                              # slot 5: stamped right before the first LDS-DMA of the prologue
            pre = (t[:, 5] - t[:, 0]) / 100.0
            print(f"    (set-up before the first LDS-DMA goes out: mean {pre.mean():.2f} us, p10 {np.percentile(pre, 10):.2f}, p90 {np.percentile(pre, 90):.2f})")
        us = us[:, :5]                                       # (the H2-output epilogue returns after stamp 4)
        names = ["prologue", "k-loop", "epilogue(LDS)", "slot stores"]
    else:
        us = us[:, :6]
        names = ["prologue", "k-loop", "epilogue(LDS)", "store issue", "store drain"]
    d = np.diff(us, axis=1)
    last = us.shape[1] - 1
    print(f"{kind} {ci}->{co}@{res} n={n}: {t.shape[0]} workgroups, kernel {plain:.3f} ms (instrumented {e0.elapsed_time(e1):.3f} ms), "
          f"span {us[:, last].max() / 1e3:.3f} ms, workgroup mean {(us[:, last] - us[:, 0]).mean():.1f} us")
    for i, nm in enumerate(names):
        print(f"    {nm:14s} mean {d[:, i].mean():7.2f} us   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")
    # how many workgroups are in each phase at a time (sampled)
    grid = np.linspace(0, us[:, last].max(), 400)
    occ = np.zeros((len(names), len(grid)))
    for i in range(len(names)):
        occ[i] = ((us[:, i, None] <= grid[None]) & (grid[None] < us[:, i + 1, None])).sum(0)
    print("    mean workgroups in phase:", {nm: round(float(occ[i].mean()), 1) for i, nm in enumerate(names)})
    # gaps between a CU slot finishing and the next workgroup starting cannot be seen directly; estimate idle from totals
    if True:
        raw = ts.cpu().numpy()[: t.shape[0]]
        dma, bar, loop = raw[:, 6].astype(np.float64), (raw[:, 7] & 0xffffffff).astype(np.float64), (raw[:, 7] >> 32).astype(np.float64)
        if dma.mean() == 0 and bar.mean() == 0:
            # (the up=2 kernel carries no in-loop stamps: their branches made the compiler sink matrix instructions past the barrier)
            print(f"    wave 0 inside the k-loop (shader cycles): total {loop.mean():.0f}; s_memtime ticks per us {loop.mean() / d[:, 1].mean():.0f}")
        else:
          print(f"    wave 0 inside the k-loop (shader cycles): total {loop.mean():.0f}, parked on vmcnt {dma.mean():.0f} "
              f"({dma.mean() / loop.mean() * 100:.1f}%), parked on the barrier {bar.mean():.0f} ({bar.mean() / loop.mean() * 100:.1f}%); "
              f"s_memtime ticks per us {loop.mean() / d[:, 1].mean():.0f}")
    if kind == "up2":
        raw3 = ts.cpu().numpy()[: t.shape[0], 3]
        w_, f_, s_ = (raw3 & 0x1fffff).astype(np.float64), ((raw3 >> 21) & 0x1fffff).astype(np.float64), ((raw3 >> 42) & 0x1fffff).astype(np.float64)
        tot = (w_ + f_ + s_).mean()
        print(f"    epilogue of wave 0 (s_memtime ticks, 4 rounds): phases->LDS + sync {w_.mean():.0f} ({w_.mean() / tot * 100:.0f}%), "
              f"FIR/activation/convert + sync {f_.mean():.0f} ({f_.mean() / tot * 100:.0f}%), slot stores {s_.mean():.0f} ({s_.mean() / tot * 100:.0f}%)")
    busy = (us[:, last] - us[:, 0]).sum()
    print(f"    sum of workgroup times / (256 CUs x span) = {busy / (256 * us[:, last].max()):.2f}")


if __name__ == "__main__":
    n = int(os.environ.get("NB_PHASE_N", "32"))
    if os.environ.get("NB_PHASE_ONLY") != "up2":
        run("up1", n, 64, 64, 256)
        run("up1", n, 128, 128, 128)
    if os.environ.get("NB_PHASE_ONLY") != "up1":
        run("up2", n, 128, 64, 256)
        run("up2", n, 384, 128, 128)
