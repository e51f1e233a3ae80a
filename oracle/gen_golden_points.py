#!/usr/bin/env python
"""Generate tests/golden/points_*.npz by running the REFERENCE's point-cloud gather in this container.

    python oracle/gen_golden_points.py            # needs /root/reference (not available on the GPU box)

The reference module /root/reference/point_cloud imports ``numba`` (not installed, not installable offline).
Harness shim, no edits to the reference: a stub ``numba`` module whose ``jit`` / ``njit`` decorators return the
function unchanged, so the decorated loops run as the plain Python/numpy they are written in (same float64
arithmetic, just slow).  ``read_and_prep_points`` (preprocess.py:45-106) is called as is: the synthetic sweep is
written to a temporary KITTI-style ``velodyne/<seq>/<frame>.bin`` first.

Each fixture stores the inputs (points, calibration, detections), the plane equations the reference derives
(float64; LAPACK results may differ in the last bit on another host, so tests take the planes from here when
they check bit-exact point membership) and the reference outputs (points, points_split).  The oracle
restatement (oracle/points_ref.py) is checked against the reference in the same run.
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def _install_numba_shim():
    nb = types.ModuleType('numba')

    def _ident(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = _ident
    nb.njit = _ident
    sys.modules['numba'] = nb


# KITTI tracking calibration of sequence 0000 (public dataset constants), padded to 4x4 like the reference's
# info['calib/*'] entries
P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
               [0.0, 0.0, 1.0, 0.002745884], [0.0, 0.0, 0.0, 1.0]])
R0 = np.array([[0.9999239, 0.00983776, -0.007445048, 0.0], [-0.009869795, 0.9999421, -0.004278459, 0.0],
               [0.007402527, 0.004351614, 0.9999631, 0.0], [0.0, 0.0, 0.0, 1.0]])
TR = np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766], [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
               [0.9998621, 0.00752379, 0.01480755, -0.2717806], [0.0, 0.0, 0.0, 1.0]])
IMG_SHAPE = np.array([375, 1242])


def make_scene(seed, n_pts, n_det, empty_boxes=1):
    """Synthetic sweep + detections.  Boxes are placed in lidar space and converted to the camera-frame KITTI
    fields the reference consumes; every box gets a cluster of points, ``empty_boxes`` of them get none."""
    rng = np.random.default_rng(seed)
    from oracle import points_ref as O
    pts = np.stack([rng.uniform(0, 70, n_pts), rng.uniform(-30, 30, n_pts), rng.uniform(-2.5, 1.0, n_pts),
                    rng.uniform(0, 1, n_pts)], 1)
    loc, dims, rot, bbox = [], [], [], []
    for i in range(n_det):
        c = np.array([rng.uniform(6, 45), rng.uniform(-8, 8), rng.uniform(-1.9, -1.2)])  # lidar, bottom centre
        wlh = np.array([rng.uniform(1.4, 2.0), rng.uniform(3.2, 4.8), rng.uniform(1.3, 1.8)])
        ry = rng.uniform(-np.pi, np.pi)
        if i >= empty_boxes:  # a cluster inside the box
            k = int(rng.integers(5, 400))
            local = (rng.uniform(-0.49, 0.49, (k, 3)) * wlh)
            local[:, 2] += wlh[2] / 2
            cs, sn = np.cos(ry), np.sin(ry)
            xy = local[:, :2] @ np.array([[cs, sn], [-sn, cs]])  # rotation_3d_in_axis(axis=2) convention
            cl = np.concatenate([xy, local[:, 2:3]], 1) + c
            pts = np.concatenate([pts, np.concatenate([cl, rng.uniform(0, 1, (k, 1))], 1)], 0)
        else:
            c = np.array([rng.uniform(60, 69), rng.uniform(25, 29), 5.0])  # above every point
        cam = (R0 @ TR @ np.append(c, 1.0))[:3]
        loc.append(cam)
        dims.append([wlh[1], wlh[2], wlh[0]])  # l, h, w
        rot.append(ry)
        # a plausible 2D box: projection of the 3D box centre +- extent
        uvw = P2[:3] @ np.append(cam, 1.0)
        u, v = uvw[0] / uvw[2], uvw[1] / uvw[2]
        ext = 721.5 * 2.0 / max(cam[2], 1.0)
        bbox.append([u - ext, v - ext * 0.8, u + ext, v + ext * 0.2])
    rng.shuffle(pts)
    dets = {'location': np.asarray(loc), 'dimensions': np.asarray(dims), 'rotation_y': np.asarray(rot),
            'bbox': np.asarray(bbox)}
    return pts.astype(np.float32), dets


CASES = [
    # name            seed  points dets  use_frustum det_type without_reflectivity
    ('points_rbbox_a', 11, 6000, 7, False, '3D', True),
    ('points_rbbox_b', 12, 20000, 12, False, '3D', False),
    ('points_frustum', 13, 8000, 6, True, '3D', True),
    ('points_det2d', 14, 5000, 5, False, '2D', True),
    ('points_tiny', 15, 300, 3, False, '3D', True),
]


def main():
    _install_numba_shim()
    sys.path.insert(0, '/root/reference')
    from point_cloud import box_np_ops as B
    from point_cloud.geometry import surface_equ_3d
    from point_cloud.preprocess import read_and_prep_points
    from oracle import points_ref as O

    out_dir = os.path.join(ROOT, 'tests', 'golden')
    worst = 0
    for name, seed, n_pts, n_det, use_frustum, det_type, wo_refl in CASES:
        pts, dets = make_scene(seed, n_pts, n_det)
        info = {'calib/R0_rect': R0, 'calib/Tr_velo_to_cam': TR, 'calib/P2': P2, 'img_shape': IMG_SHAPE}
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, 'velodyne', '0000'))
            pts.tofile(os.path.join(tmp, 'velodyne', '0000', '000000.bin'))
            ref = read_and_prep_points(info, tmp, '0000-000000.bin', dets, use_frustum=use_frustum,
                                       num_point_features=4, without_reflectivity=wo_refl, det_type=det_type,
                                       shift_bbox=dets['bbox'])
        ref_pts = np.asarray(ref['points'])
        ref_split = np.asarray(ref['points_split'], dtype=np.int64)
        # plane equations as the reference derives them (stored: see module docstring)
        rect, Trv2c, P2f = R0.astype(np.float32), TR.astype(np.float32), P2.astype(np.float32)
        C, R, T = B.projection_matrix_to_CRT_kitti(P2f)
        fr = B.get_frustum([0, 0, IMG_SHAPE[1], IMG_SHAPE[0]], C)
        fr -= T
        fr = np.linalg.inv(R) @ fr.T
        fr = B.camera_to_lidar(fr.T, rect, Trv2c)
        surf = B.corner_to_surfaces_3d_jit(fr[np.newaxis, ...])
        nv, d = surface_equ_3d(surf[:, :, :3, :])
        planes_img = np.concatenate([nv, d[..., None]], -1)
        if det_type == '3D' and not use_frustum:
            boxes = np.concatenate([dets['location'], dets['dimensions'], dets['rotation_y'][..., np.newaxis]],
                                   axis=1).astype(np.float32)
            lid = B.box_camera_to_lidar(boxes, rect, Trv2c)
            corners = B.center_to_corner_box3d(lid[:, :3], lid[:, 3:6], lid[:, 6], origin=[0.5, 0.5, 0], axis=2)
            surf = B.corner_to_surfaces_3d(corners)
        else:
            frs = B.get_frustum_v2(dets['bbox'].copy(), C)
            frs -= T
            frs = np.einsum('ij, akj->aki', np.linalg.inv(R), frs)
            frs = B.camera_to_lidar(frs, rect, Trv2c)
            surf = B.corner_to_surfaces_3d_jit(frs)
        nv, d = surface_equ_3d(surf[:, :, :3, :])
        planes_box = np.concatenate([nv, d[..., None]], -1)

        # oracle vs reference (same process)
        mine = O.prep_points(pts, rect, Trv2c, P2f, IMG_SHAPE, dets, use_frustum=use_frustum,
                             without_reflectivity=wo_refl, det_type=det_type, shift_bbox=dets['bbox'])
        assert list(ref_split) == list(mine['points_split']), name
        assert np.array_equal(np.asarray(mine['points'], dtype=np.float64), ref_pts.astype(np.float64)), name
        # restated planes: relative deviation from the reference's
        if det_type == '3D' and not use_frustum:
            pb = O.rbbox_planes(O.box_camera_to_lidar(boxes, rect, Trv2c))
        else:
            pb = O.bbox_frustum_planes(dets['bbox'].copy(), rect, Trv2c, P2f)
        dev = np.abs(pb - planes_box).max() / np.abs(planes_box).max()
        worst = max(worst, dev)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), points=pts, R0=R0, TR=TR, P2=P2, img_shape=IMG_SHAPE,
                            location=dets['location'], dimensions=dets['dimensions'], rotation_y=dets['rotation_y'],
                            bbox=dets['bbox'], use_frustum=use_frustum, det_type=det_type, without_reflectivity=wo_refl,
                            planes_img=planes_img, planes_box=planes_box, ref_points=ref_pts.astype(np.float32),
                            ref_split=ref_split)
        print('%-16s in %6d pts, %2d dets -> %6d rows, split %s...  (empty boxes: %d)' % (
            name, pts.shape[0], n_det, ref_pts.shape[0], ref_split[:4].tolist(),
            int((np.diff(ref_split) == 1).sum())))
    print('oracle == reference on every case (bit-exact rows and split); worst relative plane deviation %.2e' % worst)


if __name__ == '__main__':
    main()
