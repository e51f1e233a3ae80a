#!/usr/bin/env python
"""Throughput probe: two batches in flight (two engines = two workspaces, two streams, steps alternate) against the
one-stream step loop of bench.py, cfg3 at 16 pairs per step.  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import TrackingNet  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import init_module  # noqa: E402

KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True, appear_fpn=False,
          point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2', end_mode='avg', test_mode=2,
          neg_threshold=0.2, dropblock=0, use_dropout=False, score_fusion_arch='C', affinity_op='multiply',
          softmax_mode='none')


def main():
    B, N, M, S, pts = 16, 64, 64, 128, 2048
    ins = [make_pair(N, M, S, pts, seed=1000 + i) for i in range(B)]
    samples = [([N, M], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins]
    crops = torch.cat([x[0] for x in ins]).cuda()
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).cuda()
    models = []
    for _ in range(2):
        m = TrackingNet(**KW)
        init_module(m, seed=0)
        models.append(m.eval().cuda())
    plans = [m.make_plan(samples, S) for m in models]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    K = 20
    with torch.no_grad():
        for m, p in zip(models, plans):
            for _ in range(3):
                m.forward_batch(p, crops, points)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            models[0].forward_batch(plans[0], crops, points)
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        for i in range(4):
            with torch.cuda.stream(streams[i % 2]):
                models[i % 2].forward_batch(plans[i % 2], crops, points)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % 2]):
                models[i % 2].forward_batch(plans[i % 2], crops, points)
        torch.cuda.synchronize()
        two = time.perf_counter() - t0
    print('one batch in flight : %.2f ms per step, %.1f pairs/s' % (one / K * 1e3, K * B / one))
    print('two batches in flight: %.2f ms per step, %.1f pairs/s (%+.1f %%)' % (two / K * 1e3, K * B / two, 100 * (one / two - 1)))


if __name__ == '__main__':
    main()
