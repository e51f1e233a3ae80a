#!/usr/bin/env python
"""Timing of the GroupNorm-fed row GEMM (mmmot_gemm_rows, A_NORM_RELU prologue) per kernel variant (tile kernel of
gemm_rows.hip = 1, wide kernel of gemm_wide.hip with NH = 1 / 2: variants 3 / 4) on the w_link.conv1.3 / conv1.6 shapes.

    python tools/bench_rows_gemm.py [--rows 524288] [--n 128 512] [--k 512]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.pack import hl16_weight_shift, to_hl16  # noqa: E402
from mmmot_amd.plan import RowTiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1 << 19)
    ap.add_argument('--groups', type=int, default=32)
    ap.add_argument('--k', type=int, default=512)
    ap.add_argument('--n', type=int, nargs='*', default=[128, 512])
    ap.add_argument('--ldx', type=int, default=0, help='row pitch of the input in floats (default K): the input is the last K columns of a wider buffer, like the conv1.0 half of the stacked layer')
    ap.add_argument('--sustain', type=int, default=200)
    ap.add_argument('--pair', type=int, default=0, help='A_PAIR prologue instead of A_NORM_RELU: groups of M x M pair rows '
                    '(M = this value, a multiple of 32) generated from feature rows F [G * 2M][K]; --rows is rounded to whole groups')
    ap.add_argument('--ldf', type=int, default=0, help='row pitch of F in floats (default K)')
    ap.add_argument('--variants', type=int, nargs='*', default=[1, 3, 4])
    a = ap.parse_args()
    ops = HipOps()
    K, G = a.k, a.groups
    counts = [a.rows // G] * G
    if a.pair:
        G = max(a.rows // (a.pair * a.pair), 1)
        counts = [a.pair * a.pair] * G
    tiles = RowTiles(counts, 'cuda')
    g = torch.Generator().manual_seed(0)
    ldx = max(a.ldx, K)
    X = torch.randn(sum(counts), ldx, generator=g).cuda()[:, ldx - K:]
    sc, sh = torch.ones(G, K).cuda(), torch.zeros(G, K).cuda()
    kw = dict(X=X, sc=sc, sh=sh, amode=1)
    if a.pair:
        import numpy as np
        m, ldf = a.pair, max(a.ldf, K)
        F = torch.randn(G * 2 * m, ldf, generator=g).cuda()[:, :K]
        put = lambda v: torch.from_numpy(np.asarray(v, np.int32)).cuda()
        pair = dict(row0=tiles.g_row0, M=put([m] * G), aoff=put([2 * m * i for i in range(G)]),
                    boff=put([2 * m * i + m for i in range(G)]), uniform32=(m % 32 == 0))
        kw = dict(FA=F, FB=F, pair=pair, amode=2, pairop=0)
        ldx = ldf
    for N in a.n:
        W = torch.randn(N, K, generator=g) * K ** -0.5
        shift = hl16_weight_shift(W)
        W16 = to_hl16(W.double() * 2.0 ** shift).cuda()
        bias = torch.zeros(N).cuda()
        Y = torch.empty(sum(counts), N).cuda()
        part = torch.empty(tiles.T, 2, N).cuda()
        # (round 5 also timed the wide kernel on a stage-major weight copy [K/32][N][32] and on padded row pitches - the
        # L2-channel-camping experiment, no effect: profiles/r05/exp_gemm_wide_weight_layout_and_pitch.log; that kernel
        # variant was removed again, --ldx / --ldf still set the pitches)
        ref = None
        for v in list(a.variants):
            ops.lib.mmmot_set_gemm_rows_variant(v)
            Wv = W16
            ts = []
            for r in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(Wv, tiles, N, K, bias=bias, Y=Y, part=part, w_hl16=True, oscale=2.0 ** -shift, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            ms = ts[len(ts) // 2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(a.sustain):
                ops.gemm(Wv, tiles, N, K, bias=bias, Y=Y, part=part, w_hl16=True, oscale=2.0 ** -shift, **kw)
            e1.record()
            torch.cuda.synchronize()
            sus = e0.elapsed_time(e1) / max(a.sustain, 1)
            same = ''
            if ref is None:
                ref = (v, Y.clone(), part.clone())
            else:
                same = '; bitwise == variant %d: %s' % (ref[0], torch.equal(ref[1], Y) and torch.equal(ref[2], part))
            print('%s K=%d ld=%d N=%4d rows=%d variant %d  %.3f ms  %.0f TFLOP/s-equivalent; %d calls back to back: %.3f ms each%s' % (
                'PAIR M=%d' % a.pair if a.pair else 'NORM', K, ldx, N, sum(counts), v, ms,
                2.0 * sum(counts) * N * K / ms / 1e9, a.sustain, sus, same), flush=True)
        ops.lib.mmmot_set_gemm_rows_variant(0)


if __name__ == '__main__':
    main()
