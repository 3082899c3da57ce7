"""Developer tool (CPU, numpy): is Winograd F(2x2, 3x3) accurate enough for the split-f16 modes?

Model of one up=1 layer (128 channels, 3x3, stride 1): reference = float64 direct convolution; candidates =
  direct   split-f16 products (x = xh + xl, w = wh + wl in f16; xh wh + xh wl + xl wh, fp32 accumulation) -- today's `h3`;
  winograd the same three products on TRANSFORMED operands: V = B^T d B in fp32 on the fp32 input then split, U = G g G^T in
           float64 then split, M = sum_c U V per transformed position in fp32, Y = A^T M A in fp32.
Activations: heavy-tailed (lrelu of a normal times log-normal per-channel styles), weights normal / sqrt(fan-in) with log-normal
per-channel scales (the 'trained-like' statistics of tests/golden/gen_trained_r128.npz).  Prints the relative max / rms errors."""
import numpy as np

rs = np.random.RandomState(0)
C, O, H, W = 128, 32, 34, 34
x = rs.randn(C, H, W)
x = np.where(x > 0, x, 0.2 * x) * np.sqrt(2) * np.exp(0.5 * rs.randn(C, 1, 1))
w = rs.randn(O, C, 3, 3) / np.sqrt(9 * C) * np.exp(0.3 * rs.randn(O, 1, 1, 1)) * np.exp(0.3 * rs.randn(1, C, 1, 1))
x32, w32 = x.astype(np.float32), w.astype(np.float32)


def split(a32):
    hi = a32.astype(np.float16)
    lo = (a32 - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def prod3(ah, al, bh, bl, contract):
    """xh wh + xh wl + xl wh with fp32 accumulation (einsum in float32)."""
    f = lambda p, q: np.einsum(contract, p, q, dtype=np.float32, optimize=True)
    return f(ah, bh) + f(ah, bl) + f(al, bh)


# reference (float64), valid region
ref = np.zeros((O, H - 2, W - 2))
for a in range(3):
    for b in range(3):
        ref += np.einsum("oc,chw->ohw", w[:, :, a, b], x[:, a:a + H - 2, b:b + W - 2])

# direct split-f16
xh, xl = split(x32); wh, wl = split(w32)
direct = np.zeros((O, H - 2, W - 2), np.float32)
for a in range(3):
    for b in range(3):
        direct += prod3(wh[:, :, a, b], wl[:, :, a, b], xh[:, a:a + H - 2, b:b + W - 2], xl[:, a:a + H - 2, b:b + W - 2], "oc,chw->ohw")

# Winograd F(2x2, 3x3)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
U = np.einsum("ia,ocab,jb->ocij", G, w, G).astype(np.float32)               # offline, float64 -> fp32
uh, ul = split(U)
th, tw = (H - 2) // 2, (W - 2) // 2
d = np.stack([x32[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] for i in range(th) for j in range(tw)], 1)   # [C, T, 4, 4]
V = np.einsum("ia,ctab,jb->ctij", BT, d, BT, dtype=np.float32)              # in-kernel, fp32
vh, vl = split(V)
M = prod3(uh, ul, vh, vl, "ocij,ctij->otij")                                 # 16 GEMMs over channels, fp32 accumulation
Yt = np.einsum("ia,otab,jb->otij", AT, M, AT, dtype=np.float32)             # [O, T, 2, 2]
wino = Yt.reshape(O, th, tw, 2, 2).transpose(0, 1, 3, 2, 4).reshape(O, 2 * th, 2 * tw)

scale = np.abs(ref).max()
for name, got in (("direct split-f16", direct), ("winograd split-f16", wino)):
    e = got.astype(np.float64) - ref
    print(f"{name:20s} max |err| / max |ref| = {np.abs(e).max() / scale:.2e}   rms err / rms ref = {np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")
e32 = (np.einsum("ia,otab,jb->otij", AT.astype(np.float64), np.einsum("ocij,ctij->otij", U.astype(np.float64), V.astype(np.float64)), AT.astype(np.float64))
       .reshape(O, th, tw, 2, 2).transpose(0, 1, 3, 2, 4).reshape(O, 2 * th, 2 * tw) - ref)
print(f"{'winograd, exact products':20s} max |err| / max |ref| = {np.abs(e32).max() / scale:.2e}   (fp32 transforms alone)")
