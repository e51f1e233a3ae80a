#!/usr/bin/env python
"""(measurement script, not a pytest module; lives under tests/ because it uses the oracle as checker and CPU baseline)
Throughput of the device image preparation on one MI355X, CPU oracle / Pillow beside it.

    python tests/perf_crops.py [--dets 128] [--size 224] [--steps 50]

Workload: one KITTI-sized RGB frame resident in HBM, ``--dets`` detections per step (the cfg3 frame pair has
128), crop boxes 40..300 px.  One JSON line: detections/s, the HBM roofline of the resize kernel (algorithmic
bytes = the float32 output tensor, 3*S*S*4 B per detection, plus the crop pixels once) and Pillow on one host
core (what the reference runs in its DataLoader workers)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmot_amd import crops as CR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dets', type=int, default=128)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--steps', type=int, default=50)
    a = ap.parse_args()
    from oracle import crops_ref as O  # checker / baseline only
    H, W, S, N = 375, 1242, a.size, a.dets
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    xy = rng.uniform([0, 0], [W - 300, H - 200], (N, 2))
    bb = np.concatenate([xy, xy + rng.uniform([40, 40], [300, 200], (N, 2))], 1)
    dev = torch.from_numpy(img).cuda()
    CR.crop_resize_normalize(dev, bb, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        out = CR.crop_resize_normalize(dev, bb, S)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / a.steps
    boxes = CR.boxes_of_bboxes(bb)
    crop_bytes = float(((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).sum()) * 3
    alg = N * 3 * S * S * 4 + crop_bytes
    # parity (bit-exact) + CPU baselines on a bounded sample
    k = min(N, 16)
    t1 = time.perf_counter()
    ref_u8, ref = O.crop_resize_normalize(img, bb[:k], S)
    t_oracle = (time.perf_counter() - t1) / k
    assert np.array_equal(out[:k].cpu().numpy(), ref), 'parity'
    pil = None
    try:
        from PIL import Image
        im = Image.fromarray(img)
        t1 = time.perf_counter()
        for b in boxes[:k]:
            np.asarray(im.crop(tuple(int(v) for v in b)).resize((S, S), Image.BILINEAR), dtype=np.float32)
        pil = k / (time.perf_counter() - t1)
    except ImportError:
        pass
    print(json.dumps({
        'metric': 'detections/s through crop + antialiased bilinear resize + normalise', 'value': round(a.steps * N / dt, 1),
        'unit': 'detections/s', 'n_gpus': 1, 'steps': a.steps, 'higher_is_better': True, 'dtype': 'u8 -> i32 fixed point -> f32',
        'data': 'synthetic', 'config': {'workload': '%d detections per %dx%d frame, S=%d' % (N, H, W, S)},
        'roofline': {'bound': 'hbm', 'achieved': round(alg / (ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(alg / (ms * 1e-3) / 1e9 / 8000.0, 4), 'traffic': None, 'ms_per_step': round(ms, 4)},
        'cpu_baseline': {'value': None if pil is None else round(pil, 1), 'unit': 'detections/s', 'cores': 1,
                         'kind': 'reference', 'sample': '%d detections, Pillow crop+resize on one core; numpy oracle: %.1f/s' % (k, 1.0 / t_oracle)},
        'parity': 'bit-exact vs the oracle on %d detections' % k}))


if __name__ == '__main__':
    main()
