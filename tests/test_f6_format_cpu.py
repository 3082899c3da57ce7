"""Host side of the "f6" operand format (round 5), no GPU: the e2m3 code table, round-to-nearest-even and saturation as
v_cvt_scalef32_2xpk16_fp6_f32 does them (tools/microbench/mfma_f6_check.hip), the six-bit field packing, the block-scale rule and the weight
packer's layout (include/neube_hip.h)."""
import numpy as np
import torch

from brushstroke_engine_amd import ops


def test_e2m3_grid_roundtrip_and_rounding():
    grid = torch.tensor(ops._E2M3_GRID)
    assert torch.equal(ops.e2m3_encode(grid), torch.arange(32))
    assert torch.equal(ops.e2m3_encode(-grid[1:]), torch.arange(1, 32) + 32)
    assert torch.equal(ops.e2m3_decode(torch.arange(64)), torch.cat([grid, -grid]))
    # ties go to the even mantissa, the top saturates (the values the device test feeds the converter)
    t = torch.tensor([0.0625, 0.1875, 1.9375, 0.3125, 7.75, 1e30, -9.0, 3.875, 2.125])
    assert ops.e2m3_decode(ops.e2m3_encode(t)).tolist() == [0.0, 0.25, 2.0, 0.25, 7.5, 7.5, -7.5, 4.0, 2.0]
    # nearest grid point everywhere in range
    x = torch.linspace(-7.5, 7.5, 20001)
    q = ops.e2m3_decode(ops.e2m3_encode(x))
    best = (x[:, None] - torch.cat([grid, -grid])[None]).abs().min(dim=1).values
    assert torch.allclose((x - q).abs(), best, atol=1e-6)


def test_field_packing_roundtrip():
    g = torch.Generator().manual_seed(0)
    f = torch.randint(0, 64, (7, 3, 32), generator=g)
    b = ops.pack_f6_fields(f)
    assert b.shape == (7, 3, 24) and b.dtype == torch.uint8
    assert torch.equal(ops.unpack_f6_fields(b), f)
    # little-endian bit stream: field k at bits 6k .. 6k+5
    one = torch.zeros(32, dtype=torch.int64); one[5] = 0b101011
    bits = int.from_bytes(bytes(ops.pack_f6_fields(one[None])[0].tolist()), "little")
    assert (bits >> 30) & 63 == 0b101011 and bits == 0b101011 << 30


def test_block_exponent_rule():
    m = torch.tensor([0.0, 1.0, 1.99, 2.0, 7.5, 7.9, 8.0, 255.9, 3e-5])
    e = ops.f6_block_exponent(m)
    s = torch.exp2(e)
    r = m / s
    assert e[0] == 0 and torch.all((r[1:] >= 4) & (r[1:] < 8))          # the maximum lands in [4, 8): (7.5, 8) saturates


def test_weight_packer_layout():
    g = torch.Generator().manual_seed(3)
    o, i = 40, 48
    w = torch.randn(o, i, 3, 3, generator=g) * torch.exp(torch.randn(o, i, 1, 1, generator=g))
    p = ops.pack_conv_weight_h3f6(w)
    nch, op = 3, 64
    assert list(p.shape) == [nch, 3, 3, 2, 2, op, 8]
    raw = p.contiguous().view(torch.uint8).reshape(nch, 3, 3, 2, 2, op, 16)
    hi = p[:, :, :, :, 0].float()                                        # [chunk, ky, kx, cg, o, 8]
    want_hi = torch.zeros(nch * 16, 3, 3, op)
    want_hi[:i, :, :, :o] = w.permute(1, 2, 3, 0)
    want_hi = want_hi.half().float().reshape(nch, 2, 8, 3, 3, op).permute(0, 3, 4, 1, 5, 2)
    assert torch.equal(hi, want_hi)
    by = torch.cat([raw[:, :, :, 0, 1], raw[:, :, :, 1, 1, :, :8]], dim=-1)                 # [chunk, ky, kx, o, 24]
    codes = ops.unpack_f6_fields(by).reshape(nch, 3, 3, op, 16, 2)
    sw = torch.exp2(raw[:, :, :, 1, 1, :, 8].float() - (127 - 11))                           # the byte carries Sw * 2^-11
    dec = ops.e2m3_decode(codes) * sw[..., None, None]
    wp = torch.zeros(nch * 16, 3, 3, op); wp[:i, :, :, :o] = w.permute(1, 2, 3, 0)
    a = wp.reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)[..., ops.F6_CH]
    lo = ((wp - wp.half().float()) * 2048.0).reshape(nch, 16, 3, 3, op).permute(0, 2, 3, 4, 1)[..., ops.F6_CH]
    blockmax = torch.maximum(a.abs(), lo.abs()).amax(dim=-1, keepdim=True)
    # every field within half a step of its value (a step is 1/60 of the block maximum at the top of the range, 1/8 of the scale at the bottom)
    assert float(((dec[..., 0] - a).abs() / blockmax.clamp(min=1e-30)).max()) <= 0.5 / 4 / 2 + 1e-6 + 0.5 / 8          # saturated top: <= 0.5 of 8
    assert float(((dec[..., 0] - a).abs() / sw[..., None]).max()) <= 0.5 + 1e-6
    assert float(((dec[..., 1] - lo).abs() / sw[..., None]).max()) <= 0.5 + 1e-6
    assert not raw[:, :, :, 1, 1, :, 9:].any()                            # zeros behind the scale byte
    assert sorted(ops.F6_CH) == list(range(16))
