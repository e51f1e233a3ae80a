#!/usr/bin/env python
"""Timing experiments of the fused conv1_1 + conv1_2 + pool launch (hq8 arithmetic) on one cfg3 pair (128 crops of
128x128): patch variant 0 = product, 4 = no main-loop MFMAs, 6 = no stores, 8 = no conv1_1 prologue / slices, 13 =
conv1_1 slices without their MFMAs, 14 = without their value epilogue (all but 0 give wrong results).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import _lib  # noqa: E402
_lib.LIB_PATH = _lib.build(debug=True)  # the -DMMMOT_DEBUG build carries the timing experiments (never the product library)
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import conv1_weight_shift, hl16_weight_shift, to_hl16, to_hq8_w  # noqa: E402


def main():
    ops, lib = HipOps(), _lib.load()
    L, H, W = 128, 128, 128
    g = torch.Generator().manual_seed(0)
    crops = torch.randn(L, 3, H, W, generator=g).cuda()
    w1 = torch.zeros(64, 32)
    w1[:, :27] = torch.randn(64, 27, generator=g) * (2.0 / 27) ** 0.5
    w2 = torch.randn(9, 64, 64, generator=g) * (2.0 / 576) ** 0.5
    s1, s2 = conv1_weight_shift(w1, torch.zeros(64)), hl16_weight_shift(w2)
    w1h, w2q = to_hl16(w1.double() * 2.0 ** s1).cuda(), to_hq8_w(w2.double() * 2.0 ** s2).cuda()
    w2h = to_hl16(w2.double() * 2.0 ** s2).cuda()  # the f16x3 launch (default arithmetic)
    b = torch.zeros(64).cuda()
    out = torch.empty(L * (H // 2) * (W // 2), 64).cuda()
    for name, fn, w2 in (('hq8', ops.conv1_fused_hq8, w2q), ('hl16', ops.conv1_fused_hl16, w2h)):
        res = {}
        for r in range(8):
            for v in ((0, 4, 6, 8) if name == 'hq8' else (0, 4, 8, 13, 14)):
                lib.mmmot_set_patch_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(crops, w1h, b, 2.0 ** -s1, w2, b, 2.0 ** -s2, out, L, H, W)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res.setdefault(v, []).append(e0.elapsed_time(e1))
        lib.mmmot_set_patch_variant(0)
        for v, ts in res.items():
            ts.sort()
            print('%s variant %d: %.3f ms' % (name, v, ts[len(ts) // 2]))


if __name__ == '__main__':
    main()
