#!/usr/bin/env python
"""Time the LiDAR branch (Engine.pointnet) and its Gram-statistics pass alone at cfg3 x 8 pairs.  GPU box only.

    python tools/bench_pointnet.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmot_amd import TrackingNet  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import init_module  # noqa: E402


def timed(fn, reps=8):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device('cuda', 0)
    fusion, aff, sm, N, M, S, pts, _ = bench.WORKLOADS['cfg3']
    model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch=fusion, affinity_op=aff, softmax_mode=sm))
    init_module(model, seed=0)
    model.eval().to(dev)
    eng = model.engine()
    B = 8
    ins = [make_pair(N, M, 32, pts, seed=1000 + i) for i in range(B)]
    samples = [([N, M], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins]
    plan = model.make_plan(samples, 32)
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).to(dev)
    eng.dev = dev
    cat = eng.buf('cat', plan.Lt, 1024)
    eng.pointnet(plan, points, cat)
    torch.cuda.synchronize()
    print('Engine.pointnet, %d points: %.3f ms' % (plan.P, timed(lambda: eng.pointnet(plan, points, cat))))
    eng.pn_mlp64 = not eng.pn_mlp64
    eng.pointnet(plan, points, cat)
    print('  with pn_mlp64 = %s: %.3f ms' % (eng.pn_mlp64, timed(lambda: eng.pointnet(plan, points, cat))))
    eng.pn_mlp64 = not eng.pn_mlp64
    eng.pointnet(plan, points, cat)
    # the K = 64 layers alone, both kernels (conv3 64->64 and conv4 64->128 on the live buffers)
    pn, T = eng.P['pointnet'], plan.pt_tiles
    y1 = eng.ws['pn_y1'][:plan.P * 64].view(plan.P, 64)
    s1, h1 = eng.ws['pn1_sc'][:T.G * 64].view(T.G, 64), eng.ws['pn1_sh'][:T.G * 64].view(T.G, 64)
    for i, N_ in ((2, 64), (4, 128)):
        y = torch.empty(plan.P, N_, device=dev)
        part = torch.empty(T.T, 2, N_, device=dev)
        nbytes = plan.P * 4 * (64 + N_)
        ta = timed(lambda: eng.ops.pn_mlp64(pn['w%d_h16' % i], pn['w%d_os' % i], T, N_, y1, s1, h1, pn['b%d' % i], y, part))
        tb = timed(lambda: eng.ops.gemm(pn['w%d_h16' % i], T, N_, 64, X=y1, bias=pn['b%d' % i], Y=y, part=part, sc=s1,
                                        sh=h1, amode=1, w_hl16=True, oscale=pn['w%d_os' % i]))
        print('  64->%d: pn_mlp64 %.3f ms (%.2f TB/s) | gemm_rows %.3f ms (%.2f TB/s)' % (
            N_, ta, nbytes / ta / 1e9, tb, nbytes / tb / 1e9))
    # the Gram pass alone on the conv4 output of the run above
    x, sc, sh = eng.ws['pn_y4'][:plan.P * 128].view(plan.P, 128), eng.ws['pn4_sc'], eng.ws['pn4_sh']
    GT = plan.gram_tiles
    Gp, Sp = eng.buf64('gram_G', GT.T, 128 * 128), eng.buf64('gram_S', GT.T, 128)
    sc, sh = sc[:GT.G * 128].view(GT.G, 128), sh[:GT.G * 128].view(GT.G, 128)
    print('gram_rows<128> alone: %.3f ms (%.2f TB/s)' % (
        (t := timed(lambda: eng.ops.gram_rows(x, 128, sc, sh, GT, Gp, Sp))), plan.P * 512 / t / 1e9))


if __name__ == '__main__':
    main()
