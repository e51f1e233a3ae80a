#!/usr/bin/env python
"""Timing of the A-resident PointNet GEMM (gemm_ares.hip) vs channel count: T(N) = prologue + N/128 * per-tile.

    python tools/bench_ares.py [--rows 1048576] [--k 128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16  # noqa: E402
from mmmot_amd.plan import HalfTiles, RowTiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1 << 20)
    ap.add_argument('--modes', nargs='*', default=['stats', 'colsum'])
    ap.add_argument('--k', type=int, default=128)
    ap.add_argument('--n', type=int, nargs='*', default=[128, 256, 512, 1024])
    ap.add_argument('--variants', type=int, nargs='*', default=[0], help='mmmot_set_gemm_ares_variant values to time (1 = streaming kernel, 2 = weights in registers / independent waves, 3 = K = 64 two-barrier kernel)')
    a = ap.parse_args()
    ops = HipOps()
    R, K = a.rows, a.k
    g = torch.Generator().manual_seed(0)
    X = torch.randn(R, K, generator=g).cuda()
    sc, sh = torch.ones(1, K).cuda(), torch.zeros(1, K).cuda()
    tiles = RowTiles([R], 'cuda')
    half = HalfTiles(tiles, 'cuda')
    for N in a.n:
        W = torch.randn(N, K, generator=g) * K ** -0.5
        shift = hl16_weight_shift(W)
        W16 = to_hl16(W.double() * 2.0 ** shift).cuda()
        bias = torch.zeros(N).cuda()
        part = torch.empty(half.T, 2, N).cuda()
        cs = torch.empty(half.T, N).cuda()
        osc, osh = torch.ones(1, N).cuda(), torch.zeros(1, N).cuda()
        for mode, var in [(m, v) for m in a.modes for v in a.variants]:
            ops.lib.mmmot_set_gemm_ares_variant(var)
            kw = dict(part=part) if mode == 'stats' else dict(osc=osc, osh=osh, colsum=cs)
            ts = []
            for r in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm_ares(W16, 2.0 ** -shift, tiles, N, K, X, sc, sh, bias=bias, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            ms = ts[len(ts) // 2]
            print('K=%d N=%4d %-6s variant %d  %.3f ms  %.0f TFLOP/s-equivalent' % (K, N, mode, var, ms, 2.0 * R * N * K / ms / 1e9))
        ops.lib.mmmot_set_gemm_ares_variant(0)


if __name__ == '__main__':
    main()
