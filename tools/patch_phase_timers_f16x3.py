#!/usr/bin/env python
"""Per-item phase cycle counts of the f16x3 patch kernel (instrumented variant 9 of the -DMMMOT_DEBUG build), pooled
against unpooled, on the cfg3 layer shapes at 16 pairs per launch.  GPU box only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
_lib.LIB_PATH = _lib.build(debug=True)
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16  # noqa: E402

NAMES = ['decode + issue prologue loads', 'prologue wait', 'K loop', 'accumulators -> LDS', 'encode + issue stores',
         'closing barrier']


def main():
    ops, lib = HipOps(), _lib.load()
    g = torch.Generator().manual_seed(0)
    L = 2048
    for (H, W, Cin, Cout) in [(32, 32, 256, 256), (16, 16, 512, 512)]:
        x = torch.relu(torch.randn(L * H * W, Cin, generator=g)).cuda()
        x16 = torch.empty_like(x)
        ops.hl16_pack(x, x16)
        w = torch.randn(9, Cout, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        shift = hl16_weight_shift(w)
        w16 = to_hl16(w.double() * 2.0 ** shift).cuda()
        bias = torch.zeros(Cout).cuda()
        for pool in (False, True):
            out = torch.empty(L * H * W // (4 if pool else 1), Cout).cuda()
            lib.mmmot_set_patch_variant(9)
            buf = (ctypes.c_ulonglong * 8)()
            for r in range(3):
                if r == 1:
                    torch.cuda.synchronize()
                    lib.mmmot_debug_read_patch_timers(buf, 1)
                ops.conv3x3_hl16_patch(x16, w16, bias, out, L, H, W, Cin, Cout, pool, 2.0 ** -shift)
            torch.cuda.synchronize()
            lib.mmmot_debug_read_patch_timers(buf, 1)
            lib.mmmot_set_patch_variant(0)
            n = max(buf[7], 1)
            tot = sum(buf[i] for i in range(6))
            print('%dx%d %d->%d %s: %d items, %.0f s_memtime ticks per item' % (
                H, W, Cin, Cout, 'pooled' if pool else 'unpooled', n, tot / n))
            for i in range(6):
                print('   %-32s %9.0f  %5.1f%%' % (NAMES[i], buf[i] / n, 100.0 * buf[i] / tot))


if __name__ == '__main__':
    main()
