#!/usr/bin/env python
"""Time the training forward + backward of the pairwise block (mmmot_amd/backward.py) at the sizes of
BASELINE.json's configurations (one frame pair per call).  GPU box only.

    python tools/bench_backward.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmot_amd import TrackingNet  # noqa: E402
from mmmot_amd.backward import affinity_backward, affinity_forward_train  # noqa: E402
from mmmot_amd.plan import BatchPlan  # noqa: E402
from mmmot_amd.weights import init_module  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    for name, (aff, sm, N, M, B) in {'cfg3 N=M=64': ('multiply', 'none', 64, 64, 1), 'cfg3 x 8 pairs': ('multiply', 'none', 64, 64, 8),
                                     'cfg4 N=M=128': ('minus_abs', 'dual_add', 128, 128, 1)}.items():
        model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch='C', affinity_op=aff, softmax_mode=sm))
        init_module(model, seed=0)
        model.eval().to(dev)
        eng = model.engine()
        plan = BatchPlan([([N, M], None)] * B, 32, dev, use_points=False)
        g = torch.Generator().manual_seed(0)
        F = (torch.randn(3, plan.Lt, 512, generator=g) * 0.7).to(dev)
        R = plan.pair_tiles.R
        d_link, d_ne = torch.randn(R, generator=g).to(dev), torch.randn(2, 3, plan.Lt, generator=g).to(dev)
        tf, tb = [], []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            link, new, end, tape = affinity_forward_train(eng, plan, F)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dF, grads = affinity_backward(eng, plan, F, tape, d_link, d_ne[0], d_ne[1])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if it:
                tf.append(t1 - t0)
                tb.append(t2 - t1)
        tf.sort(); tb.sort()
        print('%-16s pair rows %6d: training forward %.2f ms, backward %.2f ms' % (name, R, tf[len(tf) // 2] * 1e3, tb[len(tb) // 2] * 1e3))


if __name__ == '__main__':
    main()
