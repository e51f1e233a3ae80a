#!/usr/bin/env python
"""Per-item phase cycle counts of the hq8 patch kernel (instrumented variant 9) on the unpooled cfg3 layer shapes
(one pair per launch).  GPU box only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
_lib.LIB_PATH = _lib.build(debug=True)  # the -DMMMOT_DEBUG build carries the timing experiments (never the product library)
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hq8_w  # noqa: E402

NAMES = ['decode + issue prologue loads', 'prologue wait', 'K loop', 'accumulators -> LDS', 'encode + issue stores',
         'closing barrier']


def main():
    ops, lib = HipOps(), _lib.load()
    g = torch.Generator().manual_seed(0)
    for (L, H, W, Cin, Cout) in [(128, 64, 64, 128, 128), (128, 32, 32, 256, 256), (128, 16, 16, 512, 512)]:
        x = torch.relu(torch.randn(L * H * W, Cin, generator=g)).cuda()
        xq = torch.empty_like(x)
        ops.hq8_pack(x, xq)
        w = torch.randn(9, Cout, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        shift = hl16_weight_shift(w)
        wq = to_hq8_w(w.double() * 2.0 ** shift).cuda()
        bias, out = torch.zeros(Cout).cuda(), torch.empty(L * H * W, Cout).cuda()
        for variant in (9,):  # (round 5: the 'no global stores' variant 10 left the kernel with the other experiments)
            lib.mmmot_set_patch_variant(variant)
            buf = (ctypes.c_ulonglong * 8)()
            for r in range(3):
                if r == 1:
                    torch.cuda.synchronize()
                    lib.mmmot_debug_read_patch_timers(buf, 1)
                ops.conv3x3_hq8(xq, wq, bias, out, L, H, W, Cin, Cout, False, 2.0 ** -shift)
            torch.cuda.synchronize()
            lib.mmmot_debug_read_patch_timers(buf, 1)
            lib.mmmot_set_patch_variant(0)
            n = max(buf[7], 1)
            tot = sum(buf[i] for i in range(6))
            print('%dx%d %d->%d%s: %d items, %.0f s_memtime ticks per item' % (
                H, W, Cin, Cout, ' (no stores)' if variant == 10 else '', n, tot / n))
            for i in range(6):
                print('   %-32s %9.0f  %5.1f%%' % (NAMES[i], buf[i] / n, 100.0 * buf[i] / tot))


if __name__ == '__main__':
    main()
