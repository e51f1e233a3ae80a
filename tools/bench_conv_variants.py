#!/usr/bin/env python
"""A/B timing of the hl16 trunk kernel's inner-loop schedule variants on the cfg3 layer shapes.

    python tools/bench_conv_variants.py [--rounds 5]

Interleaved rounds in one process (variants x layers), median ms and TFLOP/s-equivalent per variant.
Runs on the GPU box only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16, to_hq8_w  # noqa: E402

LAYERS = [  # (L, H, W, Cin, Cout, pool) at cfg3 (128 crops of 128x128)
    (128, 128, 128, 64, 64, 1), (128, 64, 64, 128, 128, 1), (128, 32, 32, 256, 256, 0), (128, 16, 16, 512, 512, 0),
    (128, 8, 8, 512, 512, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--crops', type=int, default=128, help='crops per launch (128 = one cfg3 pair)')
    ap.add_argument('--variants', default='1,11', help='1 = 128-row tiles, 4 = 256-row tiles, 7..10 = LDS-DMA kernel variants 0..3, 11 = LDS-patch kernel, 12..17 = its timing experiments 1..6 (wrong results), 21 = hq8 arithmetic (f16q8 trunk), 24/25/27 = its experiments 3/4/6')
    args = ap.parse_args()
    ops = HipOps()
    lib = _lib.load()
    variants = [int(v) for v in args.variants.split(',')]
    g = torch.Generator().manual_seed(0)
    res = {}
    for (L, H, W, Cin, Cout, pool) in [(args.crops,) + l[1:] for l in LAYERS]:
        x = torch.relu(torch.randn(L * H * W, Cin, generator=g)).cuda()
        x16 = torch.empty_like(x)
        ops.hl16_pack(x, x16)
        w = torch.randn(9, Cout, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        shift = hl16_weight_shift(w)
        w16 = to_hl16(w.double() * 2.0 ** shift).cuda()
        xq8, wq8 = torch.empty_like(x), to_hq8_w(w.double() * 2.0 ** shift).cuda()
        ops.hq8_pack(x, xq8)
        bias = torch.zeros(Cout).cuda()
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty(L * Ho * Wo, Cout).cuda()
        flops = 2.0 * L * H * W * 9 * Cin * Cout
        ref = None
        for r in range(args.rounds + 1):
            for v in variants:
                lib.mmmot_set_conv_variant(v if v < 7 else 0)
                lib.mmmot_set_dma_variant(v - 7 if 7 <= v <= 10 else 0)  # 7..10 = DMA kernel variants 0..3
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                out.fill_(float('nan'))
                e0.record()
                if v >= 21:  # hq8 arithmetic of the patch kernel (different results by design)
                    lib.mmmot_set_patch_variant(v - 21)
                    ops.conv3x3_hq8(xq8, wq8, bias, out, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
                elif v >= 11:  # LDS-resident patch kernel (12..15: timing experiments, unpooled 128-channel tiles only)
                    lib.mmmot_set_patch_variant(v - 11)
                    ops.conv3x3_hl16_patch(x16, w16, bias, out, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
                elif v >= 7:  # LDS-DMA producer/consumer kernel (its own entry point)
                    ops.conv3x3_hl16_dma(x16, w16, bias, out, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
                else:
                    ops.conv3x3_hl16(x16, w16, bias, out, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
                e1.record()
                torch.cuda.synchronize()
                if r == 0:  # warm-up round doubles as an identity check between variants
                    if ref is None:
                        ref = out.clone()
                    else:
                        if v == 10 or v >= 12:  # (and the hq8 variants: other arithmetic)
                            pass  # ASKIP timing experiment: results are wrong by construction
                        elif v >= 7:  # different K order (32-channel slabs): fp32 rounding differs, values must not
                            a, b = torch.empty_like(out), torch.empty_like(out)
                            ops.hl16_unpack(ref, a)
                            ops.hl16_unpack(out, b)
                            d = (a - b).abs().max().item()
                            assert d <= 1e-5 * max(a.abs().max().item(), 1.0), 'variant 7 differs by %g' % d
                        else:
                            assert torch.equal(ref, out), 'variant %d differs' % v
                else:
                    res.setdefault((v, (L, H, W, Cin, Cout, pool)), []).append((e0.elapsed_time(e1), flops))
    # phase timers of the instrumented variant (3) on the 32x32 256->256 layer
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    lib.mmmot_debug_read_phase_timers(buf, 1)
    lib.mmmot_set_patch_variant(0)
    lib.mmmot_set_conv_variant(3)
    L, H, W, Cin, Cout, pool = LAYERS[2]
    L = args.crops
    x16 = torch.zeros(L * H * W, Cin).cuda()
    w16 = torch.zeros(9, Cout, Cin).cuda()
    out = torch.empty(L * H * W, Cout).cuda()
    ops.conv3x3_hl16(x16, w16, torch.zeros(Cout).cuda(), out, L, H, W, Cin, Cout, False, 1.0)
    torch.cuda.synchronize()
    lib.mmmot_debug_read_phase_timers(buf, 1)
    n = max(buf[5], 1)
    names = ['vmcnt-wait + ds_write', 'barrier 1', 'issue next loads', 'ds_read + MFMA', 'barrier 2']
    tot = sum(buf[i] for i in range(5))
    print('phase cycles per (wave, stage) on 32x32 256->256 (48 MFMA = 1536 cycles of matrix pipe):')
    for i in range(5):
        print('  %-24s %8.0f  %5.1f%%' % (names[i], buf[i] / n, 100.0 * buf[i] / tot))
    lib.mmmot_set_conv_variant(0)
    print('%-8s' % 'variant' + ''.join('%22s' % ('%dx%d %d->%d%s' % (l[1], l[2], l[3], l[4], ' P' if l[5] else '')) for l in LAYERS))
    for v in variants:
        row = '%-8d' % v
        for l in [(args.crops,) + l[1:] for l in LAYERS]:
            ts = sorted(t for t, _ in res[(v, l)])
            med = ts[len(ts) // 2]
            row += '%12.3f ms %6.1f' % (med, res[(v, l)][0][1] / (med * 1e-3) / 1e12)
        print(row)


if __name__ == '__main__':
    main()
