"""Race screen: the counted-vmcnt / single-barrier pipelines of the trunk and PointNet kernels must give bitwise
identical outputs on repeated launches with the same inputs (a DMA landing late shows up as a rare wrong tile)."""
import pytest
import torch

from mmmot_amd import TrackingNet
from mmmot_amd.synth import make_pair
from mmmot_amd.weights import init_module

pytestmark = pytest.mark.gpu
KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True, appear_fpn=False,
          point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2', end_mode='avg', test_mode=2,
          neg_threshold=0.2, dropblock=0, use_dropout=False, score_fusion_arch='C', affinity_op='multiply',
          softmax_mode='none')


@pytest.mark.parametrize('trunk', ['f16x3', 'f16q8'])
@pytest.mark.parametrize('N,M,S,pts,B,reps', [(64, 64, 128, 2048, 2, 12), (7, 9, 64, 300, 3, 40)])
def test_repeated_forwards_are_bitwise_identical(N, M, S, pts, B, reps, trunk):
    model = TrackingNet(**KW)
    init_module(model, seed=0)
    model.eval().cuda()
    model.set_trunk(trunk)
    ins = [make_pair(N, M, S, pts, seed=700 + i, ragged=(pts < 1000)) for i in range(B)]
    samples = [([N, M], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins]
    plan = model.make_plan(samples, S)
    crops = torch.cat([x[0] for x in ins]).cuda()
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).cuda()
    first = None
    for r in range(reps):
        out = model.engine().forward(plan, crops, points)
        cur = [out[k].clone() for k in ('det', 'link', 'new', 'end', 'cat')]
        if first is None:
            first = cur
            assert all(torch.isfinite(t).all() for t in cur)
        else:
            for a, b, k in zip(first, cur, ('det', 'link', 'new', 'end', 'cat')):
                assert torch.equal(a, b), 'run %d differs in %s' % (r, k)


def test_two_stream_forward_equals_one_stream_and_is_capturable():
    """Engine.two_streams (LiDAR branch on a side stream, fork / join by events): same bits as the one-stream forward,
    repeatedly, and inside a hipGraph capture."""
    model = TrackingNet(**KW)
    init_module(model, seed=0)
    model.eval().cuda()
    N, M, S, pts = 9, 12, 64, 300
    ins = [make_pair(N, M, S, pts, seed=730 + i, ragged=True) for i in range(2)]
    samples = [([N, M], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins]
    plan = model.make_plan(samples, S)
    crops = torch.cat([x[0] for x in ins]).cuda()
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).cuda()
    eng = model.engine()
    keys = ('det', 'link', 'new', 'end', 'cat')
    eng.two_streams = False
    ref = [eng.forward(plan, crops, points)[k].clone() for k in keys]
    eng.two_streams = True
    for r in range(10):
        out = eng.forward(plan, crops, points)
        for a, k in zip(ref, keys):
            assert torch.equal(a, out[k]), 'two-stream run %d differs in %s' % (r, k)
    graphed = model.capture(plan, crops, points)
    for r in range(3):
        res = graphed(crops, points)
    torch.cuda.synchronize()
    one = model.forward_batch(plan, crops, points)
    for (d0, l0, n0, e0), (d1, l1, n1, e1) in zip(res, one):
        assert torch.equal(d0, d1) and torch.equal(n0, n1) and torch.equal(e0, e1)
        assert all(torch.equal(a, b) for a, b in zip(l0, l1))
