#!/usr/bin/env python
"""Time one training step of the WHOLE network (reference tracking_model.py:50-66: training-mode forward -> TrackingLoss ->
backward -> SGD step) at the shape of BASELINE.json configs[0] (one KITTI-like frame pair: N=10, M=12, 224x224 crops,
ragged ~300 pts/det) and at a cfg2-like sample (N=M=32, 64x64 crops, 512 pts/det).  GPU box only.

    python tools/bench_train.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmot_amd import TrackingLoss, TrackingNet  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import init_module  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    for name, (N, M, S, pts, ragged) in {'cfg1 N=10 M=12 224x224': (10, 12, 224, 300, True),
                                         'cfg2-like N=M=32 64x64': (32, 32, 64, 512, False)}.items():
        model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch='C', affinity_op='multiply', softmax_mode='none'))
        init_module(model, seed=0)
        model.to(dev).train()
        crit = TrackingLoss(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
        opt = torch.optim.SGD(model.parameters(), lr=1e-4)
        dets, info, ds = make_pair(N, M, S, pts, seed=4000, ragged=ragged)
        dets, info = dets.to(dev), {k: v.to(dev) for k, v in info.items()}
        g = torch.Generator().manual_seed(1)
        L = N + M
        gt_det = (torch.rand(L, generator=g) > 0.3).float().to(dev)
        gt_new, gt_end = (torch.rand(L, generator=g) > 0.6).float().to(dev), (torch.rand(L, generator=g) > 0.6).float().to(dev)
        gt_link = [(torch.rand(1, N, M, generator=g) > 0.9).float().to(dev)]
        for frozen in (False, True):
            model.freeze_appearance = frozen
            ts, tf, losses = [], [], []
            for it in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                det, links, new, end, trans = model(dets, info, ds)
                loss = crit(ds, gt_det, gt_link, gt_new, gt_end, det, links, new, end, trans)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                opt.zero_grad()
                loss.backward()
                opt.step()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                losses.append(loss.item())
                if it:
                    ts.append(t2 - t0)
                    tf.append(t1 - t0)
            ts.sort(); tf.sort()
            print('%-24s %-22s step %.1f ms (forward + loss %.1f ms), loss %.4f -> %.4f, peak memory %.2f GB' % (
                name, 'image branch frozen' if frozen else 'whole network', ts[len(ts) // 2] * 1e3, tf[len(tf) // 2] * 1e3,
                losses[0], losses[-1], torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == '__main__':
    main()
