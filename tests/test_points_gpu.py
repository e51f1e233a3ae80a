"""Point-cloud gather on the GPU (csrc/points_gather.hip through the C-ABI) against the reference's outputs
(golden fixtures), against the CPU oracle on fresh inputs, and through size-independent properties at 1M points."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import points_ref as O
from mmmot_amd import points as PT
from test_points_oracle import GOLD, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_prep_points_matches_reference(path):
    z, info, dets, kw = load(path)
    got = PT.prep_points(torch.from_numpy(z['points']).cuda(), info, dets, shift_bbox=dets['bbox'], **kw)
    assert list(got['points_split']) == z['ref_split'].tolist()
    assert got['points'].is_cuda
    assert np.array_equal(got['points'].cpu().numpy(), z['ref_points'])  # bit-exact, same order, zero rows for empty boxes


@pytest.mark.parametrize('P,N,F,drop', [(1, 1, 4, False), (255, 3, 3, False), (256, 2, 4, True), (1000, 300, 4, True),
                                        (70001, 17, 4, False)])
def test_gather_vs_oracle_fresh_inputs(P, N, F, drop):
    """Random axis-aligned-ish boxes over random points: ragged sizes, > 256 polygons (chunked calls), F = 3."""
    rng = np.random.default_rng(P + N)
    pts = rng.uniform(-10, 10, (P, F)).astype(np.float32)
    boxes = np.concatenate([rng.uniform(-8, 8, (N, 3)), rng.uniform(0.5, 6, (N, 3)), rng.uniform(-3, 3, (N, 1))], 1)
    planes = O.rbbox_planes(boxes)
    rows, split = O.gather_per_box(pts, planes)
    if drop and F == 4:
        rows = rows[:, :3]
    got, gsplit = PT.gather_points(torch.from_numpy(pts).cuda(), planes, pad_empty=True, drop_reflectivity=drop)
    assert gsplit.tolist() == split.tolist()
    assert np.array_equal(got.cpu().numpy(), rows.astype(np.float32))


def test_gather_properties_at_one_million_points():
    """Sizes the oracle's dense [P, N, 6] tensor would not like: check the defining properties instead - every
    emitted row is inside its polygon, counts equal an independent device-side count, input order is kept."""
    P, N = 1 << 20, 64
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(P, 3, generator=g) * torch.tensor([70.0, 60.0, 3.5]) + torch.tensor([0.0, -30.0, -2.5]))
    idx = torch.arange(P, dtype=torch.float32).view(P, 1)  # exact in fp32 below 2^24: order witness
    pts = torch.cat([xyz, idx], 1).cuda()
    rng = np.random.default_rng(6)
    boxes = np.concatenate([rng.uniform([5, -25, -2], [60, 25, -1], (N, 3)), rng.uniform(2, 8, (N, 3)),
                            rng.uniform(-3, 3, (N, 1))], 1)
    planes = O.rbbox_planes(boxes)
    rows, split = PT.gather_points(pts, planes, pad_empty=True)
    pl = torch.from_numpy(planes).cuda()
    x = pts[:, :3].double()
    sign = ((x[:, None, None, 0] * pl[None, :, :, 0] + x[:, None, None, 1] * pl[None, :, :, 1]) +
            x[:, None, None, 2] * pl[None, :, :, 2]) + pl[None, :, :, 3]
    inside = ~(sign >= 0).any(-1)                       # [P, N] on the device, same float64 expression
    counts = inside.sum(0).cpu().numpy()
    assert np.array_equal(np.diff(split), np.maximum(counts, 1))
    assert counts.sum() > 10000
    for j in range(N):
        r = rows[split[j]:split[j + 1]]
        if counts[j] == 0:
            assert r.shape[0] == 1 and float(r.abs().sum()) == 0.0
            continue
        ids = r[:, 3].long()
        assert bool((ids[1:] > ids[:-1]).all())                              # input order kept
        assert bool(inside[ids, j].all())                                    # only inside points
        assert torch.equal(r, pts[ids])                                      # rows copied verbatim


def test_batched_gather_equals_per_frame_reference():
    """All golden frames in ONE launch sequence (image frustum fused into the box test) == the reference's
    two-stage result per frame, bit for bit."""
    cases = [load(p) for p in GOLD if not bool(np.load(p)['use_frustum']) and str(np.load(p)['det_type']) == '3D']
    assert len(cases) >= 2
    sweeps = [torch.from_numpy(z['points']).cuda() for z, _, _, _ in cases]
    res = PT.prep_points_batched(sweeps, [i for _, i, _, _ in cases], [d for _, _, d, _ in cases],
                                 without_reflectivity=False)
    for (z, info, dets, kw), r in zip(cases, res):
        ref = z['ref_points'] if not kw['without_reflectivity'] else None
        assert r['points_split'] == z['ref_split'].tolist()
        got = r['points'].cpu().numpy()
        if ref is not None:
            assert np.array_equal(got, ref)
        else:  # the fixture dropped the reflectivity column
            assert np.array_equal(got[:, :3], z['ref_points'])


def test_batched_gather_frustum_path_and_many_sweeps():
    z, info, dets, kw = load([p for p in GOLD if 'frustum' in p][0])
    n = 9
    sweeps = [torch.from_numpy(z['points']).cuda() for _ in range(n)]
    res = PT.prep_points_batched(sweeps, [info] * n, [dets] * n, use_frustum=True, without_reflectivity=True,
                                 shift_bboxes=[dets['bbox']] * n)
    for r in res:
        assert r['points_split'] == z['ref_split'].tolist()
        assert np.array_equal(r['points'].cpu().numpy(), z['ref_points'])


@pytest.mark.parametrize('filtered', [False, True], ids=['no_filter', 'filter'])
def test_batched_gather_many_polygons_per_sweep_vs_oracle(filtered):
    """Sweeps with 1, 16, 17, 70 and 256 polygons (one and many 16-polygon membership passes, more than one
    64-polygon scatter pass), ragged point counts incl. one point and a block boundary, boxes that hold a large
    share of a sweep (long LDS queues), F = 3 and 4: rows and split == the oracle's per-box gather, bit for bit."""
    rng = np.random.default_rng(77 + filtered)
    for F in (3, 4):
        npts = [1, 256, 257, 3000, 1500]
        npoly = [1, 16, 17, 70, 256]
        sweeps, planes, exp_rows, exp_split = [], [], [], [0]
        filt_planes = []
        for P, N in zip(npts, npoly):
            pts = rng.uniform(-10, 10, (P, F)).astype(np.float32)
            boxes = np.concatenate([rng.uniform(-8, 8, (N, 3)), rng.uniform(0.5, 6, (N, 3)), rng.uniform(-3, 3, (N, 1))], 1)
            boxes[0, 3:6] = 30.0  # one box that swallows most of the sweep
            pl = O.rbbox_planes(boxes)
            src = pts
            if filtered:  # the filter polygon: a big box of its own; emitted rows must be inside both
                fbox = np.array([[1.0, -1.0, 0.0, 14.0, 16.0, 18.0, 0.3]])
                fpl = O.rbbox_planes(fbox)
                filt_planes.append(fpl)
                keep = pts[O.inside_planes(pts, fpl)[:, 0]]
                src = keep
            if len(src):
                rows, split = O.gather_per_box(src, pl)
            else:  # nothing survives the filter: every polygon owns its one zero row
                rows, split = np.zeros((N, F), np.float32), np.arange(N + 1)
            exp_rows.append(rows.astype(np.float32))
            exp_split.extend((np.asarray(split[1:]) + exp_split[-1]).tolist())
            sweeps.append(pts)
            planes.append(pl)
        rows0 = np.concatenate([[0], np.cumsum(npts)])
        allpl = np.concatenate(planes + filt_planes)
        filters = [sum(npoly) + i for i in range(len(npts))] if filtered else None
        got, gsplit = PT.gather_points_batched(torch.from_numpy(np.concatenate(sweeps)).cuda(), rows0, allpl, npoly,
                                               filters=filters, pad_empty=True)
        assert gsplit.tolist() == exp_split
        assert np.array_equal(got.cpu().numpy(), np.concatenate(exp_rows))
