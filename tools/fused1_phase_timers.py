#!/usr/bin/env python
"""Per-tile phase cycle counts (thread 0 = a K-loop wave) of the fused conv1_1 + conv1_2 + pool launch in f16x3 arithmetic
(instrumented variant 9) on one cfg3 pair.  GPU box only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
_lib.LIB_PATH = _lib.build(debug=True)
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import conv1_weight_shift, hl16_weight_shift, to_hl16  # noqa: E402

NAMES = ['-', 'tile top (first tile: serial conv1_1)', 'K loop (18 stages) + range guard', 'accumulators -> LDS',
         'encode + issue stores', 'closing barrier']


def main():
    ops, lib = HipOps(), _lib.load()
    L, H, W = 128, 128, 128
    g = torch.Generator().manual_seed(0)
    crops = torch.randn(L, 3, H, W, generator=g).cuda()
    w1 = torch.zeros(64, 32)
    w1[:, :27] = torch.randn(64, 27, generator=g) * (2.0 / 27) ** 0.5
    w2 = torch.randn(9, 64, 64, generator=g) * (2.0 / 576) ** 0.5
    s1, s2 = conv1_weight_shift(w1, torch.zeros(64)), hl16_weight_shift(w2)
    w1h, w2h = to_hl16(w1.double() * 2.0 ** s1).cuda(), to_hl16(w2.double() * 2.0 ** s2).cuda()
    b = torch.zeros(64).cuda()
    out = torch.empty(L * (H // 2) * (W // 2), 64).cuda()
    lib.mmmot_set_patch_variant(9)
    buf = (ctypes.c_ulonglong * 8)()
    for r in range(3):
        if r == 1:
            torch.cuda.synchronize()
            lib.mmmot_debug_read_patch_timers(buf, 1)
        ops.conv1_fused_hl16(crops, w1h, b, 2.0 ** -s1, w2h, b, 2.0 ** -s2, out, L, H, W)
    torch.cuda.synchronize()
    lib.mmmot_debug_read_patch_timers(buf, 1)
    lib.mmmot_set_patch_variant(0)
    v = [int(x) for x in buf]
    items = max(1, v[7])
    tot = sum(v[1:6])
    print('fused conv1 launch, %d tile visits of thread 0 over 2 launches (s_memtime ticks)' % items)
    for i in range(1, 6):
        print('  %-42s %9.1f ticks/tile  %5.1f %%' % (NAMES[i], v[i] / items, 100.0 * v[i] / max(1, tot)))


if __name__ == '__main__':
    main()
