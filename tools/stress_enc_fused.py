"""Race hunt for the fused encoder launch (nb_enc_stem_conv3x3_f8: the stem's slabs are written into the buffers the K loop has just left, by
waves that are not barrier-aligned with the readers of the OTHER buffer): the same launch many times on the same inputs, every result compared
bit for bit with the first, on full and under-filled chips and with other work on a second stream.
    gpurun -- 'python tools/stress_enc_fused.py'"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import _lib, encoder as encmod
dev = torch.device("cuda:0")
lib = _lib.lib()
P = lambda t: t.data_ptr()
rs = np.random.RandomState(0)
total = bad = 0
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
for (n, h, w, co, reps) in [(32, 256, 256, 128, 300), (8, 128, 128, 128, 300), (3, 64, 192, 144, 300), (1, 16, 64, 48, 300)]:
    x = torch.from_numpy(rs.rand(n, 1, h, w).astype(np.float32)).to(dev)
    w50 = np.zeros([64, 50], np.float32); w50[:, :49] = (rs.randn(64, 49) / 7).astype(np.float32)
    w50d, b0 = torch.from_numpy(w50).to(dev), torch.from_numpy(rs.randn(64).astype(np.float32)).to(dev)
    w1 = torch.from_numpy(encmod.pack_enc_weight_f8((rs.randn(co, 64, 3, 3) / 24).astype(np.float32))).to(dev)
    b1 = torch.from_numpy(rs.randn(co).astype(np.float32)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    ys = []
    for r in range(reps):
        y = torch.empty([n, co // 8, 2, h // 2, w // 2, 8], dtype=torch.float16, device=dev)
        if r % 3 == 1:
            with torch.cuda.stream(side):
                junk2 = junk @ junk                 # something else on the chip
        _lib.check(lib.nb_enc_stem_conv3x3_f8(P(x), P(w50d), P(b0), 0, P(w1), P(b1), P(y), 1, n, h, w, co, 0.01, st), "fused")
        ys.append(y)
        if len(ys) == 50 or r == reps - 1:
            torch.cuda.synchronize()
            if r < 50:
                first = ys[0].clone()
            for y_ in ys:
                total += 1
                if not torch.equal(y_.view(torch.int16), first.view(torch.int16)):
                    bad += 1
            ys = []
    print(f"n={n} {h}x{w} c_out={co}: {reps} launches, differing so far {bad}")
print(f"{total} launches, {bad} differ from the first of their shape")
