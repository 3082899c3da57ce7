"""Would fp8 (e4m3, fixed scales) be enough for the two CORRECTION products of the split-f16 scheme?
x*w ~= xh*wh [f16 x f16]  +  fp8(xl*2^9)*fp8(wh) * 2^-9  +  fp8(xh/4)*fp8(wl*2^11) * 2^-9     (mode 'f8')
End-to-end pixel error of the R=128 generator against the fp32 oracle (CPU simulation; the GPU kernel does not exist)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
torch.set_num_threads(8)
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from oracle import neube_oracle as orc


def f8(t):
    """round to fp8 e4m3fn (saturating) and back"""
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def split(t):
    hi = t.half().float(); lo = (t - hi).half().float(); return hi, lo


class SplitOracle(orc.OracleGenerator):
    mode = "h3"

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
        if self.mode == "h3":
            y = y + conv(xl, wh) + conv(xh, wl)
        elif self.mode == "f8":
            y = y + conv(f8(xl * 512), f8(wh)) / 512 + conv(f8(xh / 4), f8(wl * 2048)) / 512
        self.maxabs = max(getattr(self, "maxabs", 0.0), float(xm.abs().max()))
        y = y * d.reshape(n, -1, 1, 1) + noise
        return orc.bias_act(y, sd[f"{name}.bias"], act="lrelu", gain=orc.SQRT2, clamp=self.cfg.conv_clamp)


for res in (128,):
    cfg = cfgmod.style1_config(res)
    sd = wmod.random_state_dict(cfg, 0)
    n = 2
    z = synthetic.batch_z(cfg, n, 594); geom = synthetic.geom_features(cfg, n, 0); pos = synthetic.positions(cfg, n, 0)
    img32, d32 = orc.OracleGenerator(cfg, sd)(z, None, geom, positions=pos, return_debug_data=True)
    for mode in ("h3", "f8"):
        S = SplitOracle(cfg, sd); S.mode = mode
        img, dd = S(z, None, geom, positions=pos, return_debug_data=True)
        print(res, mode, "vs fp32 oracle: uvs", float((dd["uvs"] - d32["uvs"]).abs().max()), "img", float((img - img32).abs().max()),
              "max |x*style|", S.maxabs)
