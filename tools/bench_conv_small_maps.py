#!/usr/bin/env python
"""A/B timing of the trunk kernel on feature maps of at most 4 x 4 pixels (conv5_x at 64-pixel crops, BASELINE cfg2 /
cfg4): the whole-map geometry (16 maps per 256-row tile, no halo; round 4) against the haloed 8 x 8 geometry those layers
ran in before (one map per 8 x 8 block: 25 % tile fill), selected with the test knob mmmot_set_patch_min_block.

    python tools/bench_conv_small_maps.py [--crops 2048] [--rounds 7] [--q8]

Interleaved rounds in one process; median ms and TFLOP/s-equivalent (algorithmic FLOPs of the layer).  GPU box only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16, to_hq8_w  # noqa: E402

LAYERS = [  # (H, W, Cin, Cout, pool)
    (4, 4, 512, 512, 0), (4, 4, 512, 512, 1), (2, 2, 512, 512, 0), (8, 8, 512, 512, 0), (8, 8, 256, 512, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--crops', type=int, default=2048, help='crops per launch (2048 = one cfg2 step, 8192 = one cfg4 step)')
    ap.add_argument('--q8', action='store_true', help='hq8 arithmetic (f16q8 trunk)')
    args = ap.parse_args()
    ops = HipOps()
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    L = args.crops
    print('%d crops per launch, %s arithmetic' % (L, 'hq8' if args.q8 else 'hl16'))
    for (H, W, Cin, Cout, pool) in LAYERS:
        x = torch.relu(torch.randn(L * H * W, Cin, generator=g)).cuda()
        xr = torch.empty_like(x)
        (ops.hq8_pack if args.q8 else ops.hl16_pack)(x, xr)
        w = torch.randn(9, Cout, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        shift = hl16_weight_shift(w)
        wr = (to_hq8_w if args.q8 else to_hl16)(w.double() * 2.0 ** shift).cuda()
        bias = torch.zeros(Cout).cuda()
        osc = torch.full((Cout,), 2.0 ** -shift).cuda()
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        outs = {}
        flops = 2.0 * L * H * W * 9 * Cin * Cout
        ts = {0: [], 8: []}
        for r in range(args.rounds + 1):
            for mb in (0, 8):
                assert lib.mmmot_set_patch_min_block(mb) == 0
                out = torch.empty(L * Ho * Wo, Cout).cuda()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                (ops.conv3x3_hq8 if args.q8 else ops.conv3x3_hl16_patch)(xr, wr, bias, out, L, H, W, Cin, Cout, bool(pool), osc)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts[mb].append(e0.elapsed_time(e1))
                outs[mb] = out
        lib.mmmot_set_patch_min_block(0)
        same = torch.equal(outs[0].view(torch.int32), outs[8].view(torch.int32))
        row = '%dx%d %d->%d%s' % (H, W, Cin, Cout, ' P' if pool else '  ')
        for mb in (0, 8):
            med = sorted(ts[mb])[len(ts[mb]) // 2]
            row += '   min_block=%d: %8.3f ms %6.1f TF-eq' % (mb, med, flops / (med * 1e-3) / 1e12)
        print(row + '   bitwise equal: %s' % same)


if __name__ == '__main__':
    main()
