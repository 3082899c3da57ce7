"""Race hunt for the persistent workgroups of the two large conv kernels (round 6): the four large launches of the BASELINE step, repeated -- alone
and with another large launch running beside them on a second stream --, every result compared bit for bit with the one-workgroup-per-tile
launch's.  A prefetch that lands on LDS somebody still reads (or a table rewritten under a late epilogue) would show as a sporadic difference.
    gpurun -- 'python tools/stress_persistent.py'      (NB_REPS=300)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, ops
lib = _lib.lib()
for f in ("nb_debug_set_up1_persistent", "nb_debug_set_up2v_persistent", "nb_debug_set_persistent_wgs_per_cu"):
    getattr(lib, f).argtypes, getattr(lib, f).restype = [ctypes.c_int], None
reps = int(os.environ.get("NB_REPS", "300"))
side = torch.cuda.Stream()
layers = []
for up, ci, co, res, n in ((1, 128, 128, 128, 32), (1, 64, 64, 256, 32), (2, 384, 128, 128, 32), (2, 128, 64, 256, 32)):
    rs = np.random.RandomState(ci + up)
    hin = res // up
    x = torch.from_numpy(rs.randn(n, ci, hin, hin).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(9 * ci)).astype(np.float32)).cuda()
    st = torch.from_numpy(rs.uniform(0.5, 1.5, (n, ci)).astype(np.float32)).cuda()
    L = dict(up=up, ci=ci, co=co, res=res, n=n, hin=hin, xh=ops.pack_h2f8(x, st), wp=ops.pack_conv_weight_h3f8(w),
             nst=torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda(), dco=torch.from_numpy(rs.uniform(0.5, 1.5, (n, co)).astype(np.float32)).cuda(),
             bias=torch.from_numpy(rs.randn(co).astype(np.float32)).cuda(), noise=torch.from_numpy(rs.randn(n, res, res).astype(np.float32)).cuda())
    del x
    L["out"] = [torch.zeros(ops.h2_shape(n, co, res, res), dtype=torch.int16, device="cuda") for _ in range(3)]
    layers.append(L)


def run(L, out, stream):
    a = (L["xh"].data_ptr(), L["ci"], L["wp"].data_ptr(), L["dco"].data_ptr(), L["noise"].data_ptr(), L["res"] * L["res"], L["bias"].data_ptr(), None, out.data_ptr(),
         L["nst"].data_ptr(), L["co"], L["co"])
    tail = (L["n"], L["hin"], L["hin"], L["co"], 0.2, 1.4142135, 256.0, stream.cuda_stream)
    rc = lib.nb_modconv3x3_up1_h3_ex(*a, None, 1, 1, *tail) if L["up"] == 1 else lib.nb_modconv3x3_up2_h3_ex(*a, 1, 1, *tail)
    _lib.check(rc, "conv")


main = torch.cuda.current_stream()
for L in layers:                                         # references: one workgroup per tile
    lib.nb_debug_set_up1_persistent(0); lib.nb_debug_set_up2v_persistent(0)
    run(L, L["out"][0], main)
torch.cuda.synchronize()
lib.nb_debug_set_up1_persistent(1); lib.nb_debug_set_up2v_persistent(1)          # (the up=1 form is opt-in since the end of round 6)
bad = 0
for k in (0, 1):                                         # default (4 workgroups per CU), then every workgroup resident (1 per CU)
    lib.nb_debug_set_persistent_wgs_per_cu(k)
    for i, L in enumerate(layers):
        other = layers[(i + 1) % len(layers)]
        nb = 0
        for rep in range(reps):
            L["out"][1].zero_()
            if rep % 2:                                  # every other repetition with a neighbour on the side stream
                side.wait_stream(main)
                run(other, other["out"][2], side)
            run(L, L["out"][1], main)
            torch.cuda.synchronize()
            if not torch.equal(L["out"][0], L["out"][1]):
                nb += 1
        bad += nb
        print(f"up{L['up']} {L['ci']}->{L['co']}@{L['res']} wgs/cu {'4 (default)' if k == 0 else 1}: {reps} persistent launches, {nb} differ from the one-workgroup-per-tile result", flush=True)
lib.nb_debug_set_persistent_wgs_per_cu(0)
lib.nb_debug_set_up1_persistent(-1); lib.nb_debug_set_up2v_persistent(-1)
print("STRESS", "FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
