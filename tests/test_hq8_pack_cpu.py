"""Host side of the hq8 record format (include/mmmot_hip.h) - no GPU needed:
byte layout of activation / weight records, saturation, decode error, and the arithmetic identity the kernel relies on
(hi*hi + 2^-3 (a8*w_lo8 + a_lo8*w8) ~ a*w) through the torch statement of the C-ABI contract (tests/fake_ops.py)."""
import numpy as np
import torch

from fake_ops import TorchOps
from mmmot_amd.pack import from_hq8_act, hl16_weight_shift, hq8_parts, to_hl16, to_hq8_act, to_hq8_w


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def e4m3_decode(codes):
    return torch.from_numpy(np.asarray(codes, dtype=np.uint8)).view(torch.float8_e4m3fn).float()


def test_activation_record_layout_and_decode():
    x = torch.relu(rnd(5, 64, seed=1)) * 7.0
    rec = to_hq8_act(x)
    assert rec.dtype == torch.float32 and rec.shape == x.shape          # same bytes as fp32: buffers are shared
    raw = rec.contiguous().view(torch.uint8).reshape(5, 2, 128)          # one 128-byte record per 32 channels
    hi = raw[:, :, :64].contiguous().view(torch.float16).reshape(5, 64).float()
    a8 = e4m3_decode(raw[:, :, 64:96].contiguous().numpy()).reshape(5, 64)
    l8 = e4m3_decode(raw[:, :, 96:128].contiguous().numpy()).reshape(5, 64)
    assert torch.equal(hi, x.half().float())                              # [0,64): fp16 hi, channel order
    assert torch.equal(a8, (x * 0.25).to(torch.float8_e4m3fn).float())    # [64,96): e4m3(a / 4)
    assert torch.equal(l8, ((x.double() - hi.double()) * 512.0).float().to(torch.float8_e4m3fn).float())
    dec = from_hq8_act(rec)
    assert torch.equal(dec, (hi.double() + l8.double() / 512.0).float())
    assert ((dec - x).abs() <= x.abs() * 2.0 ** -15 + 2.0 ** -20).all()   # fp16 half-ulp (2^-12) times e4m3's 2^-4
    parts = hq8_parts(rec)
    assert torch.equal(parts[0].float(), hi) and torch.equal(parts[1].float(), a8) and torch.equal(parts[2].float(), l8)


def test_activation_record_saturates_instead_of_overflowing():
    x = torch.tensor([[1792.0, 1793.0, 5000.0, 65000.0, 1e9, -1e9, 0.0, -0.0] * 4])
    rec = to_hq8_act(x)
    hi, a8, l8 = hq8_parts(rec)
    assert torch.isfinite(hi).all() and torch.isfinite(a8).all() and torch.isfinite(l8).all()
    assert float(a8.max()) == 448.0 and float(a8.min()) == -448.0         # e4m3 range, never NaN
    assert float(hi.max()) <= 65504.0
    # above 1792 the e4m3 copies saturate: the decoded value degrades to fp16 class, not worse
    dec = from_hq8_act(rec)
    ok = x.abs() <= 65000.0
    assert ((dec - x).abs()[ok] <= x.abs()[ok] * 2.0 ** -11).all()


def test_weight_record_layout():
    w = rnd(9 * 64, 64, seed=2, scale=0.05)
    shift = hl16_weight_shift(w)
    ws = w.double() * 2.0 ** shift
    assert 8192.0 <= float(ws.abs().max()) < 16384.0 * 1.0001              # max |w'| next to 2^14
    rec = to_hq8_w(ws)
    hi, wl8, w8 = hq8_parts(rec)
    assert torch.equal(hi.float(), ws.half().float())
    assert torch.equal(wl8.float(), ((ws - hi) * 32.0).float().to(torch.float8_e4m3fn).float())   # e4m3(32 w_lo)
    assert torch.equal(w8.float(), (hi / 64.0).float().to(torch.float8_e4m3fn).float())           # e4m3(w_hi / 64)
    assert float(w8.abs().max()) <= 256.0 and float(wl8.abs().max()) <= 448.0
    # the hl16 copy of the same weights carries the same hi halves: the two arithmetics share the main term
    C = ws.shape[-1]
    hl_hi = to_hl16(ws).contiguous().view(torch.float16).reshape(ws.shape[0], C // 8, 2, 8)[:, :, 0, :].reshape(ws.shape[0], C)
    assert torch.equal(hl_hi.float(), hi.float())


def test_hq8_product_identity():
    """hi*hi + 2^-3 (a8 * w_lo8 + a_lo8 * w8) reproduces a*w to ~2^-15 per product (vs 2^-11 for the main term alone)"""
    L, H, W, Cin, Cout = 2, 8, 8, 64, 64
    x = torch.relu(rnd(L * H * W, Cin, seed=3)) * 3.0
    w = rnd(9, Cout, Cin, seed=4, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = torch.zeros(Cout)
    shift = hl16_weight_shift(w)
    emu = TorchOps(torch.float64)
    got = emu._conv_hq8(hq8_parts(to_hq8_act(x)), to_hq8_w(w.double() * 2.0 ** shift), bias, L, H, W, Cin, Cout,
                        False, 2.0 ** -shift)
    exact = torch.zeros(L * H * W, Cout)
    emu.conv3x3(x.view(L, H, W, Cin), w, bias, exact, L, H, W, Cin, Cout, False, False)
    main_only = torch.zeros(L * H * W, Cout)
    emu.conv3x3(x.half().float().view(L, H, W, Cin), w.half().float(), bias, main_only, L, H, W, Cin, Cout, False, False)
    scale = float(exact.abs().max())
    err_q8 = float((got.float() - exact).abs().max()) / scale
    err_main = float((main_only - exact).abs().max()) / scale
    assert err_q8 < 3e-5, err_q8
    assert err_main > 4 * err_q8, (err_main, err_q8)   # the correction terms do their job


def test_per_channel_weight_scales_keep_low_gain_channels_precise():
    """Output channels with gains spread over 1e6 (a trained, BatchNorm-folded layer).  With ONE power-of-two scale
    for the layer the e4m3 copies of the low-gain channels underflow (their correction terms vanish: the effective
    weight hi + e4m3(32 w_lo) / 32 is fp16-class, 2^-12); with a scale per output channel (pack.hl16_channel_shifts)
    every channel's effective weight is within 2^-15 of the true one, relative to that channel's largest weight."""
    from mmmot_amd.pack import hl16_channel_shifts
    Cout, Cin = 128, 64
    g = torch.Generator().manual_seed(5)
    gain = torch.pow(10.0, torch.rand(Cout, generator=g) * 6.0 - 4.0).double()
    w = rnd(9, Cout, Cin, seed=6, scale=0.05).double() * gain.view(1, -1, 1)

    def per_channel_error(scaled, unscale):
        hi, wl8, _ = hq8_parts(to_hq8_w(scaled))
        eff = (hi + wl8 / 32.0) * unscale
        return ((eff - w).abs().amax(dim=(0, 2)) / w.abs().amax(dim=(0, 2))).max().item()

    sh = hl16_channel_shifts(w)
    assert int(sh.max() - sh.min()) >= 15
    e_vec = per_channel_error(w * torch.pow(2.0, sh.double()).view(1, -1, 1), torch.pow(2.0, -sh.double()).view(1, -1, 1))
    s1 = hl16_weight_shift(w)
    e_one = per_channel_error(w * 2.0 ** s1, 2.0 ** -s1)
    assert e_vec < 2.0 ** -15, e_vec
    assert e_one > 2.0 ** -13, e_one
    # the a_lo * w_hi operand: e4m3(w_hi / 64) keeps its 3 mantissa bits for every channel only with per-channel scales
    _, _, w8 = hq8_parts(to_hq8_w(w * torch.pow(2.0, sh.double()).view(1, -1, 1)))
    assert (w8.abs().amax(dim=(0, 2)) >= 128.0).all()   # every channel's largest weight sits in e4m3's top binades
    _, _, w8 = hq8_parts(to_hq8_w(w * 2.0 ** s1))
    assert (w8.abs().amax(dim=(0, 2)) < 2.0 ** -6).any()  # one scale: some channels are entirely subnormal / zero
