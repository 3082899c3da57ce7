"""Would fp6 (e2m3) be enough for the two CORRECTION products of the split-f16 scheme?
  f8:  x*w ~= xh*wh [f16 x f16] + e4m3(xl 2^9) e4m3(w) 2^-9 + e4m3(x/4) e4m3(wl 2^11) 2^-9          (fixed scales)
  f6:  x*w ~= xh*wh [f16 x f16] + Sx Sw 2^-11 [ e2m3(xl 2^11 / Sx) e2m3(w / Sw) + e2m3(x / Sx) e2m3(wl 2^11 / Sw) ]
       Sx = one E8M0 scale per (pixel, 16-channel chunk), Sw = one per (c_out, tap, 16-channel chunk): e2m3 has 3 mantissa bits like
       e4m3 but only a 2^6 range, so the block maximum is placed at the top of it (the MFMA applies one scale per lane = per 32
       K values = [16 channels x {xl, x}] of one pixel and tap).
End-to-end pixel error of the R=128 generator against the fp32 oracle on the three weight sets the GPU tests use (random,
trained-like, hdr).  CPU simulation of the arithmetic only (the convolution itself runs in float64)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
torch.set_num_threads(8)
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from oracle import neube_oracle as orc


def f8(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def e2m3(t):
    """round-to-nearest-even onto the e2m3 grid (max 7.5, subnormal step 0.125), saturating"""
    a = t.abs().clamp(max=7.5)
    step = torch.where(a < 2, 0.125, torch.where(a < 4, 0.25, 0.5))
    return torch.sign(t) * torch.round(a / step) * step


def block_scale(m, exact=True):
    """E8M0 scale for a block whose largest magnitude is m: the power of two that puts m into (3.75, 7.5] (exact) or into [4, 8)
    with saturation of (7.5, 8) (the cheaper rule: exponent of the maximum minus 2)."""
    m = m.clamp(min=2.0 ** -40)
    e = torch.floor(torch.log2(m))
    if exact:
        e = torch.where(m / torch.exp2(e - 2) > 7.5, e + 1, e)
    return torch.exp2(e - 2)


def split(t):
    hi = t.half().float(); lo = (t - hi).half().float(); return hi, lo


class SplitOracle(orc.OracleGenerator):
    mode = "h3"
    exact_scale = True
    min_res = 32           # layers below run hi/lo f16 (the small-tile kernels) in every mode

    def layer(self, spec, x, w, norm_noise_positions=None, input_noise=None, fused_modconv=True, taps=None):
        sd, name = self.sd, spec.name
        styles = orc.fully_connected(w, sd[f"{name}.affine.weight"], sd[f"{name}.affine.bias"])
        noise_const = sd[f"{name}.noise_const"]
        if norm_noise_positions is not None:
            noise_const = orc.shifted_const_noise(noise_const, sd[f"{name}.noise_grid"], norm_noise_positions)
        noise = noise_const * sd[f"{name}.noise_strength"]
        W = sd[f"{name}.weight"]
        n = x.shape[0]
        d = ((W.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        xm = x * styles.reshape(n, -1, 1, 1)
        xh, xl = split(xm); wh, wl = split(W)

        def conv(a, b):
            return orc.conv2d_resample(a.double(), b.double(), f=self.filter.double(), up=spec.up, padding=1, flip_weight=(spec.up == 1)).float()
        y = conv(xh, wh)
        mode = self.mode if (spec.in_res >= self.min_res and xm.shape[1] % 16 == 0) else "h3"
        if mode == "h3":
            y = y + conv(xl, wh) + conv(xh, wl)
        elif mode == "f8":
            y = y + conv(f8(xl * 512), f8(wh)) / 512 + conv(f8(xh / 4), f8(wl * 2048)) / 512
        elif mode == "f6":
            N, C, H, Wd = xm.shape
            xb = xm.reshape(N, C // 16, 16, H, Wd)
            Sx = block_scale(xb.abs().amax(dim=2, keepdim=True), self.exact_scale)
            q_xl = (e2m3(xl.reshape_as(xb) * 2048 / Sx) * Sx).reshape_as(xm)
            q_x = (e2m3(xb / Sx) * Sx).reshape_as(xm)
            O, I, kh, kw = W.shape
            wb = W.reshape(O, I // 16, 16, kh, kw)
            Sw = block_scale(wb.abs().amax(dim=2, keepdim=True), self.exact_scale)
            q_w = (e2m3(wb / Sw) * Sw).reshape_as(W)
            q_wl = (e2m3(wl.reshape_as(wb) * 2048 / Sw) * Sw).reshape_as(W)
            y = y + (conv(q_xl, q_w) + conv(q_x, q_wl)) / 2048
        y = y * d.reshape(n, -1, 1, 1) + noise
        return orc.bias_act(y, sd[f"{name}.bias"], act="lrelu", gain=orc.SQRT2, clamp=self.cfg.conv_clamp)


def run(label, cfg, sd, z, geom, pos):
    img32, d32 = orc.OracleGenerator(cfg, sd)(z, None, geom, positions=pos, return_debug_data=True)
    for mode, exact in (("h3", True), ("f8", True), ("f6", True), ("f6", False)):
        S = SplitOracle(cfg, sd); S.mode = mode; S.exact_scale = exact
        img, dd = S(z, None, geom, positions=pos, return_debug_data=True)
        print(f"{label:12s} {mode}{'' if exact else ' (scale = exponent - 2, saturating)':38s} uvs {float((dd['uvs'] - d32['uvs']).abs().max()):.2e}  "
              f"img {float((img - img32).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    cfg = cfgmod.style1_config(128)
    n = 2
    z = synthetic.batch_z(cfg, n, 1234); geom = synthetic.geom_features(cfg, n, 3)
    pos = np.array([[37, 211], [4095, 17]], np.int64)
    run("random", cfg, wmod.random_state_dict(cfg, 0), z, geom, pos)
    run("trained-like", cfg, wmod.trained_like_state_dict(cfg, 0), z, geom, pos)
    run("hdr", cfg, wmod.hdr_state_dict(cfg, 0), z, geom, pos)
