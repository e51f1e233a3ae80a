#!/usr/bin/env python
"""Timing of the trunk's f16x3 weight-gradient launch (mmmot_conv3x3_wgrad_f16) at the layer shapes of a cfg1 training step
(22 crops of 224 x 224).  GPU box only.

    python tools/bench_wgrad.py [--layers conv3 conv4] [--reps 6]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd.ops import HipOps  # noqa: E402
from mmmot_amd.train_vgg import _wgrad_shares  # noqa: E402

SHAPES = {'conv1_2': (224, 64, 64), 'conv2_1': (112, 64, 128), 'conv2_2': (112, 128, 128), 'conv3_1': (56, 128, 256),
          'conv3': (56, 256, 256), 'conv4_1': (28, 256, 512), 'conv4': (28, 512, 512), 'conv5': (14, 512, 512)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', nargs='*', default=list(SHAPES))
    ap.add_argument('--crops', type=int, default=22)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--shares', type=int, default=0, help='0 = the training step\'s choice')
    a = ap.parse_args()
    ops = HipOps()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for name in a.layers:
        S, cin, cout = SHAPES[name]
        L, rows = a.crops, a.crops * S * S
        dZ = (torch.randn(rows, cout, generator=g) * 1e-4).to(dev)
        A = torch.relu(torch.randn(rows, cin, generator=g)).to(dev)
        ns = a.shares or _wgrad_shares(dev, cin, cout, rows, True)
        dW = torch.empty(ns, 9 * cout * cin, device=dev)
        amax = dZ.abs().max().reshape(1)
        ts = []
        for r in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv3x3_wgrad(dZ, A, L, S, S, cin, cout, ns, dW, amax=amax)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        fl = 2.0 * rows * 9 * cin * cout
        print('%-8s %4d x %3d -> %3d  rows %8d  shares %3d  %.3f ms  %.0f TFLOP/s (of 833 f16x3-equivalent: %.2f)' %
              (name, S, cin, cout, rows, ns, ms, fl / ms / 1e9, fl / ms / 1e9 / 833.3))


if __name__ == '__main__':
    main()
