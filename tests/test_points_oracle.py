"""Point-cloud gather: the CPU oracle against the reference's own outputs, and the product's
host-side plane construction against the reference's planes - no GPU needed.

Fixtures tests/golden/points_*.npz come from oracle/gen_golden_points.py, which runs the reference's
``read_and_prep_points`` (numba decorators shimmed to the identity) in the build container."""
import glob
import os

import numpy as np
import pytest

from oracle import points_ref as O
from mmmot_amd import points as PT

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'points_*.npz')))


def load(path):
    z = np.load(path)
    info = {'calib/R0_rect': z['R0'], 'calib/Tr_velo_to_cam': z['TR'], 'calib/P2': z['P2'], 'img_shape': z['img_shape']}
    dets = {'location': z['location'], 'dimensions': z['dimensions'], 'rotation_y': z['rotation_y'], 'bbox': z['bbox']}
    kw = dict(use_frustum=bool(z['use_frustum']), det_type=str(z['det_type']),
              without_reflectivity=bool(z['without_reflectivity']))
    return z, info, dets, kw


def test_fixtures_present():
    assert len(GOLD) >= 5


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_reference(path):
    z, info, dets, kw = load(path)
    f32 = lambda k: info[k].astype(np.float32)
    got = O.prep_points(z['points'], f32('calib/R0_rect'), f32('calib/Tr_velo_to_cam'), f32('calib/P2'),
                        info['img_shape'], dets, shift_bbox=dets['bbox'], **kw)
    assert list(got['points_split']) == z['ref_split'].tolist()
    assert np.array_equal(np.asarray(got['points'], dtype=np.float32), z['ref_points'])  # bit-exact rows, same order


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_membership_from_the_reference_planes_is_exact(path):
    """With the plane equations the reference derived (stored), the restated float64 test reproduces the
    reference's gather bit for bit - the decision does not depend on this host's LAPACK."""
    z, info, dets, kw = load(path)
    keep = O.inside_planes(z['points'], z['planes_img'])[:, 0]
    rows, split = O.gather_per_box(z['points'][keep], z['planes_box'])
    if kw['without_reflectivity']:
        rows = rows[:, :3]
    assert split.tolist() == z['ref_split'].tolist()
    assert np.array_equal(rows.astype(np.float32), z['ref_points'])


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_product_host_planes_match_the_reference(path):
    z, info, dets, kw = load(path)
    rect, Tr, P2 = (info[k].astype(np.float32) for k in ('calib/R0_rect', 'calib/Tr_velo_to_cam', 'calib/P2'))
    img = PT.image_frustum_planes(rect, Tr, P2, info['img_shape'])
    if kw['det_type'] == '3D' and not kw['use_frustum']:
        boxes = np.concatenate([dets['location'], dets['dimensions'], dets['rotation_y'][..., None]], 1).astype(np.float32)
        box = PT.rbbox_planes(boxes, rect, Tr)
    else:
        box = PT.bbox_frustum_planes(dets['bbox'].copy(), rect, Tr, P2)
    for mine, ref in ((img, z['planes_img']), (box, z['planes_box'])):
        assert mine.dtype == np.float64 and mine.shape == ref.shape
        assert np.abs(mine - ref).max() <= 1e-12 * np.abs(ref).max()


def test_gather_refuses_host_tensors():
    import torch
    with pytest.raises(RuntimeError):
        PT.gather_points(torch.zeros(4, 4), np.zeros((1, 6, 4)))
