#!/usr/bin/env python
"""Timing of the Gram statistics pass (csrc/gram.hip): gram_rows over R rows of K = 128 channels in super-tiles of 4096.

    python tools/bench_gram.py [--rows 4194304] [--groups 16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.plan import RowTiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1 << 22)
    ap.add_argument('--groups', type=int, default=16)
    ap.add_argument('--k', type=int, default=128)
    ap.add_argument('--variants', type=int, nargs='*', default=[1, 2, 0], help='K = 128: mmmot_set_gram128_variant values to time')
    a = ap.parse_args()
    ops = HipOps()
    K, G = a.k, a.groups
    counts = [a.rows // G] * G
    tiles = RowTiles(counts, 'cuda', tile=4096)
    X = torch.randn(sum(counts), K, generator=torch.Generator().manual_seed(0)).cuda()
    sc, sh = torch.ones(G, K).cuda(), torch.zeros(G, K).cuda()
    Gp = torch.zeros(tiles.T, K * K, dtype=torch.float64).cuda()
    Sp = torch.zeros(tiles.T, K, dtype=torch.float64).cuda()
    ref = None
    for v in (a.variants if K == 128 else [0]):
        if K == 128:
            ops.lib.mmmot_set_gram128_variant(v)
        Gp.zero_()
        Sp.zero_()
        ts = []
        for r in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gram_rows(X, K, sc, sh, tiles, Gp, Sp)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        same = ''
        if ref is None:
            ref = (Gp.clone(), Sp.clone())
        else:
            same = '  bitwise == first variant: G %s, S %s' % (torch.equal(ref[0], Gp), torch.equal(ref[1], Sp))
        print('gram_rows K=%d rows=%d super-tiles=%d variant %d: %.3f ms  %.2f TB/s  checksum %.6e%s' % (
            K, sum(counts), tiles.T, v, ms, sum(counts) * K * 4 / ms / 1e9, float(Gp.sum()), same))
    if K == 128:
        ops.lib.mmmot_set_gram128_variant(0)

if __name__ == '__main__':
    main()
