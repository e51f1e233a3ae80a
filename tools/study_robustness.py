#!/usr/bin/env python
"""CPU study (no GPU) of the trunk arithmetics on trained-like weight statistics, through the torch emulation of the
C-ABI (tests/fake_ops.py) with the engine's range guard active: per case the score error of f16q8 (with the guard's
fallbacks), of forced f16q8 (guard off) and of f16x3 against the CPU oracle, plus the guard's events.

    python tools/study_robustness.py [--quick]
"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fake_ops import TorchOps  # noqa: E402
from mmmot_amd import TrackingNet  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import calibrate_bn, generate_state_dict_trained  # noqa: E402
from oracle import restatement as R  # noqa: E402

KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True, appear_fpn=False,
          point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2', end_mode='avg', test_mode=2,
          neg_threshold=0.2, dropblock=0, use_dropout=False, score_fusion_arch='C', affinity_op='multiply',
          softmax_mode='none')


def run(profile, seed, scale, trunk, guard, N=6, M=5, S=64, pts=40):
    torch.manual_seed(0)
    m = TrackingNet(**KW)
    sd = generate_state_dict_trained(m.state_dict(), seed, profile)
    dets, info, ds = make_pair(N, M, S, pts, seed=4000 + seed, ragged=True)
    dets = dets * scale
    if profile == 'calibrated':
        calibrate_bn(sd, make_pair(8, 8, S, 4, seed=4100 + seed)[0] * scale)
    m.load_state_dict(sd)
    m.eval()
    m.set_ops(TorchOps())
    m.set_trunk(trunk)
    eng = m.engine()
    eng.range_guard = guard
    cfg = dict(fusion='C', affinity_op='multiply', softmax_mode='none', neg_threshold=0.2, score_arch='branch_cls')
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], [N, M])
        out = m(dets, info, ds)
    err = max((out[0] - ref[0]).abs().max().item(), (out[1][0] - ref[1][0]).abs().max().item(),
              (out[2] - ref[2]).abs().max().item(), (out[3] - ref[3]).abs().max().item())
    return err, eng.trunk, eng.range_events


if __name__ == '__main__':
    quick = '--quick' in sys.argv
    torch.set_num_threads(8)
    for profile in ('calibrated', 'wild'):
        for seed in ((0,) if quick else (0, 1, 2)):
            for scale in (1e-2, 1.0, 1e2, 1e3):
                e_g, t_g, ev = run(profile, seed, scale, 'f16q8', True)
                e_q, _, _ = run(profile, seed, scale, 'f16q8', False)
                e_x, _, _ = run(profile, seed, scale, 'f16x3', False)
                print('%-10s seed %d scale %-6g  guarded f16q8 -> %-5s %.2e | forced f16q8 %.2e | f16x3 %.2e | events %s' % (
                    profile, seed, scale, t_g, e_g, e_q, e_x,
                    [(e['was'], e['now'], e['e4m3_saturated'], e['fp16_clamped']) for e in ev]), flush=True)
