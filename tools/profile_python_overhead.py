#!/usr/bin/env python
"""Where does the host time of one reference-shaped call (B=1, cfg1 shape) go?  cProfile over 30 eager forwards.
GPU box only."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmot_amd import TrackingNet  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import init_module  # noqa: E402

dev = torch.device('cuda', 0)
model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch='A', affinity_op='multiply', softmax_mode='none'))
init_module(model, seed=0)
model.eval().to(dev)
ins = []
for i in range(6):
    dets, info, ds = make_pair(10, 12, 224, 300, seed=3000 + i, ragged=True)
    ins.append((dets.to(dev), {k: v.to(dev) for k, v in info.items()}, ds))
with torch.no_grad():
    for x in ins:
        model(*x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(5):
        for x in ins:
            model._plans.clear()
            model(*x)
    issue = (time.perf_counter() - t0) / 30
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 30
    print('host issue time per forward %.3f ms, wall incl. drain %.3f ms' % (issue * 1e3, total * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for r in range(5):
        for x in ins:
            model._plans.clear()
            model(*x)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
