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


def test_reference_call_with_trunk_first_and_lidar_beside_it_is_race_free():
    """TrackingNet.forward launches the trunk before the point split is back (Engine.image_first), uploads the plan tables on
    the side stream and runs the LiDAR branch there beside the trunk: 150 alternating calls of three shapes (plan-cache hits
    and misses, fresh input tensors every call) give the bits of the plain single-stream order."""
    model = TrackingNet(**{**KW, 'score_fusion_arch': 'A'})
    init_module(model, seed=0)
    model.eval().cuda()
    shapes = [(10, 12, 224, 300), (3, 17, 64, 40), (20, 2, 128, 120)]
    cases = [make_pair(N, M, S, pts, seed=760 + i, ragged=True) for i, (N, M, S, pts) in enumerate(shapes)]

    def dev(c):
        dets, info, ds = c
        return dets.cuda(), {k: v.cuda() for k, v in info.items()}, ds

    model.image_first = False
    with torch.no_grad():
        want = []
        for c in cases:
            det, links, new, end, _ = model(*dev(c))
            want.append([t.clone() for t in (det, links[0], new, end)])
        model.image_first = True
        assert model.engine().pn_beside_trunk
        for r in range(150):
            i = (r * 7 + r // 5) % 3
            if r % 11 == 0:
                model._plans.clear()  # plan-cache miss: tables built and uploaded while the trunk runs
            det, links, new, end, _ = model(*dev(cases[i]))
            for a, b, k in zip(want[i], (det, links[0], new, end), ('det', 'link', 'new', 'end')):
                assert torch.equal(a, b), 'call %d (shape %d) differs in %s' % (r, i, k)
