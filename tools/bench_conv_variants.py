#!/usr/bin/env python
"""A/B timing of the trunk kernel (conv3x3_hl16_patch.hip) and its timing experiments on the cfg3 layer shapes.

    python tools/bench_conv_variants.py [--rounds 5] [--variants 11,21]

Interleaved rounds in one process (variants x layers), median ms and TFLOP/s-equivalent per variant.
Variants: 11 = hl16 arithmetic (f16x3 trunk), 12..17 = its timing experiments 1..6 (WRONG results by construction:
no loads / no barriers / no fragment reads / no MFMAs / no epilogue / no stores), 18 = every fourth MFMA only (the
matrix-core work per streamed weight byte of a Winograd F(2x2,3x3) stage), 19 = the product kernel with
`s_setprio 1` for waves 4-7, 20 = with streaming stores in the unpooled epilogue (both: correct results), 23 = chained tiles without
their closing LDS barrier (timing experiment, correctness unverified), 21 = hq8 arithmetic (f16q8 trunk),
24 / 25 / 27 / 28 / 32 = its experiments 3 / 4 / 6 / 7 / 11.  The experiments exist only in the -DMMMOT_DEBUG build of
the library, which this tool builds and loads (libmmmot_hip_debug.so).  Runs on the GPU box only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
_lib.LIB_PATH = _lib.build(debug=True)
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16, to_hq8_w  # noqa: E402

LAYERS = [  # (L, H, W, Cin, Cout, pool) at cfg3 (128 crops of 128x128)
    (128, 64, 64, 64, 128, 0), (128, 64, 64, 128, 128, 1), (128, 32, 32, 256, 256, 0), (128, 16, 16, 512, 512, 0),
    (128, 8, 8, 512, 512, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--crops', type=int, default=128, help='crops per launch (128 = one cfg3 pair)')
    ap.add_argument('--variants', default='11,21')
    args = ap.parse_args()
    ops = HipOps()
    lib = _lib.load()
    variants = [int(v) for v in args.variants.split(',')]
    g = torch.Generator().manual_seed(0)
    res = {}
    layers = [(args.crops,) + l[1:] for l in LAYERS]
    for (L, H, W, Cin, Cout, pool) in layers:
        x = torch.relu(torch.randn(L * H * W, Cin, generator=g)).cuda()
        x16 = torch.empty_like(x)
        ops.hl16_pack(x, x16)
        w = torch.randn(9, Cout, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        shift = hl16_weight_shift(w)
        w16 = to_hl16(w.double() * 2.0 ** shift).cuda()
        xq8, wq8 = torch.empty_like(x), to_hq8_w(w.double() * 2.0 ** shift).cuda()
        ops.hq8_pack(x, xq8)
        bias = torch.zeros(Cout).cuda()
        osc = torch.full((Cout,), 2.0 ** -shift).cuda()
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty(L * Ho * Wo, Cout).cuda()
        flops = 2.0 * L * H * W * 9 * Cin * Cout
        for r in range(args.rounds + 1):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if v >= 21 and v != 23:
                    lib.mmmot_set_patch_variant(v - 21)
                    ops.conv3x3_hq8(xq8, wq8, bias, out, L, H, W, Cin, Cout, bool(pool), osc)
                else:
                    lib.mmmot_set_patch_variant({18: 12, 19: 15, 20: 16, 23: 17}.get(v, v - 11))  # 18: a quarter of the MFMAs; 19: s_setprio 1 for waves 4-7; 20: streaming stores
                    ops.conv3x3_hl16_patch(x16, w16, bias, out, L, H, W, Cin, Cout, bool(pool), osc)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res.setdefault((v, (L, H, W, Cin, Cout, pool)), []).append((e0.elapsed_time(e1), flops))
    lib.mmmot_set_patch_variant(0)
    print('%-8s' % 'variant' + ''.join('%22s' % ('%dx%d %d->%d%s' % (l[1], l[2], l[3], l[4], ' P' if l[5] else '')) for l in LAYERS))
    for v in variants:
        row = '%-8d' % v
        for l in layers:
            ts = sorted(t for t, _ in res[(v, l)])
            med = ts[len(ts) // 2]
            row += '%12.3f ms %6.1f' % (med, res[(v, l)][0][1] / (med * 1e-3) / 1e12)
        print(row)


if __name__ == '__main__':
    main()
