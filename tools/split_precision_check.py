import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
torch.set_num_threads(8)
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from oracle import neube_oracle as orc
import torch.nn.functional as F

def split(t):
    hi = t.half().float(); lo = (t - hi).half().float(); return hi, lo

class SplitOracle(orc.OracleGenerator):
    mode = 3
    def layer(self, spec, x, w, norm_noise_positions=None, input_noise=None, fused_modconv=True, taps=None):
        sd, name = self.sd, spec.name
        styles = orc.fully_connected(w, sd[f"{name}.affine.weight"], sd[f"{name}.affine.bias"])
        noise_const = sd[f"{name}.noise_const"]
        if norm_noise_positions is not None:
            noise_const = orc.shifted_const_noise(noise_const, sd[f"{name}.noise_grid"], norm_noise_positions)
        noise = noise_const * sd[f"{name}.noise_strength"]
        W = sd[f"{name}.weight"]
        n = x.shape[0]
        d = ((W.unsqueeze(0) * styles.reshape(n,1,-1,1,1)).square().sum(dim=[2,3,4]) + 1e-8).rsqrt()
        xm = x * styles.reshape(n, -1, 1, 1)
        xh, xl = split(xm); wh, wl = split(W)
        def conv(a, b):
            return orc.conv2d_resample(a.double(), b.double(), f=self.filter.double(), up=spec.up, padding=1, flip_weight=(spec.up == 1)).float()
        y = conv(xh, wh)
        if self.mode >= 2: y = y + conv(xl, wh)
        if self.mode >= 3: y = y + conv(xh, wl)
        y = y * d.reshape(n, -1, 1, 1) + noise
        return orc.bias_act(y, sd[f"{name}.bias"], act="lrelu", gain=orc.SQRT2, clamp=self.cfg.conv_clamp)

for res in (128,):
    cfg = cfgmod.style1_config(res)
    sd = wmod.random_state_dict(cfg, 0)
    n = 2
    z = synthetic.batch_z(cfg, n, 594); geom = synthetic.geom_features(cfg, n, 0); pos = synthetic.positions(cfg, n, 0)
    ref64 = orc.OracleGenerator(cfg, sd, dtype=torch.float64)
    img64, d64 = ref64(z, None, geom, positions=pos, return_debug_data=True)
    img32, d32 = orc.OracleGenerator(cfg, sd)(z, None, geom, positions=pos, return_debug_data=True)
    print(res, "fp32 oracle vs fp64: uvs", float((d32["uvs"].double()-d64["uvs"]).abs().max()), "img", float((img32.double()-img64).abs().max()))
    for mode in (1, 2, 3):
        S = SplitOracle(cfg, sd); S.mode = mode
        img, dd = S(z, None, geom, positions=pos, return_debug_data=True)
        print(res, "split mode", mode, "vs fp64: uvs", float((dd["uvs"].double()-d64["uvs"]).abs().max()), "img", float((img.double()-img64).abs().max()),
              " vs fp32 oracle img", float((img-img32).abs().max()))
