"""CPU restatement (TEST INFRASTRUCTURE ONLY - never imported by the product path) of the reference's
point-cloud gather that feeds ``det_info['points']`` / ``det_info['points_split']``:

    read_and_prep_points          /root/reference/point_cloud/preprocess.py:45-106   (minus the file read)
    remove_outside_points         /root/reference/point_cloud/box_np_ops.py:629-640
    get_frustum_points            box_np_ops.py:643-653
    points_in_rbbox               box_np_ops.py:688-699
    center_to_corner_box3d        box_np_ops.py:312-337, corners_nd :147-178, rotation_3d_in_axis :236-254
    corner_to_surfaces_3d         box_np_ops.py:702-720 (== corner_to_surfaces_3d_jit :723-743)
    surface_equ_3d                /root/reference/point_cloud/geometry.py:84-93
    _points_in_convex_polygon_3d_jit  geometry.py:96-114  (numba; restated vectorised)
    camera_to_lidar / box_camera_to_lidar / projection_matrix_to_CRT_kitti / get_frustum / get_frustum_v2

Pinned against the reference itself: oracle/gen_golden_points.py imports /root/reference/point_cloud with a
``numba`` shim whose jit decorators are the identity (numba is not installed; the decorated functions are plain
Python/numpy) and stores inputs, plane equations and outputs under tests/golden/points_*.npz.

Arithmetic notes that matter for bit-exactness of the inside/outside decision:
  * ``camera_to_lidar`` appends ``np.ones`` (float64) to float32 coordinates, so lidar boxes, corners, surfaces,
    normal vectors and d are float64 even though every input is float32;
  * the jit loop evaluates ``sign = p0*n0 + p1*n1 + p2*n2 + d`` left to right in float64 (float32 point times
    float64 normal), no FMA contraction, and a point is inside iff sign < 0 for all 6 surfaces.
"""
import numpy as np


# ---- box geometry (host-side, O(boxes)) ---------------------------------------------------------------------
def corners_nd(dims, origin=0.5):
    ndim = int(dims.shape[1])
    corners_norm = np.stack(np.unravel_index(np.arange(2 ** ndim), [2] * ndim), axis=1).astype(dims.dtype)
    if ndim == 2:
        corners_norm = corners_norm[[0, 1, 3, 2]]
    elif ndim == 3:
        corners_norm = corners_norm[[0, 1, 3, 2, 4, 5, 7, 6]]
    corners_norm = corners_norm - np.array(origin, dtype=dims.dtype)
    return dims.reshape([-1, 1, ndim]) * corners_norm.reshape([1, 2 ** ndim, ndim])


def rotation_3d_in_axis(points, angles, axis=0):
    rot_sin, rot_cos = np.sin(angles), np.cos(angles)
    ones, zeros = np.ones_like(rot_cos), np.zeros_like(rot_cos)
    if axis == 1:
        rot_mat_T = np.stack([[rot_cos, zeros, -rot_sin], [zeros, ones, zeros], [rot_sin, zeros, rot_cos]])
    elif axis == 2 or axis == -1:
        rot_mat_T = np.stack([[rot_cos, -rot_sin, zeros], [rot_sin, rot_cos, zeros], [zeros, zeros, ones]])
    elif axis == 0:
        rot_mat_T = np.stack([[zeros, rot_cos, -rot_sin], [zeros, rot_sin, rot_cos], [ones, zeros, zeros]])
    else:
        raise ValueError('axis should in range')
    return np.einsum('aij,jka->aik', points, rot_mat_T)


def center_to_corner_box3d(centers, dims, angles=None, origin=(0.5, 1.0, 0.5), axis=1):
    corners = corners_nd(dims, origin=list(origin))
    if angles is not None:
        corners = rotation_3d_in_axis(corners, angles, axis=axis)
    corners += centers.reshape([-1, 1, 3])
    return corners


_SURF_IDX = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)


def corner_to_surfaces_3d(corners):
    """[N, 8, 3] -> [N, 6, 4, 3], normals pointing inwards."""
    return corners[:, _SURF_IDX]


def surface_equ_3d(polygon_surfaces):
    surface_vec = polygon_surfaces[:, :, :2, :] - polygon_surfaces[:, :, 1:3, :]
    normal_vec = np.cross(surface_vec[:, :, 0, :], surface_vec[:, :, 1, :])
    d = np.einsum('aij, aij->ai', normal_vec, polygon_surfaces[:, :, 0, :])
    return normal_vec, -d


def planes_of_surfaces(surfaces):
    """[N, 6, 4, 3] -> float64 [N, 6, 4] rows (nx, ny, nz, d): the operand of the device kernel."""
    normal_vec, d = surface_equ_3d(surfaces[:, :, :3, :])
    return np.concatenate([normal_vec, d[..., None]], axis=-1).astype(np.float64)


def inside_planes(points, planes):
    """points [P, >=3] (float32), planes [N, 6, 4] float64 -> bool [P, N]; geometry.py:96-114."""
    p = points[:, :3]
    n = planes[..., :3]
    # ((p0*n0 + p1*n1) + p2*n2) + d, float64, no contraction
    sign = p[:, None, None, 0] * n[None, :, :, 0]
    sign = sign + p[:, None, None, 1] * n[None, :, :, 1]
    sign = sign + p[:, None, None, 2] * n[None, :, :, 2]
    sign = sign + planes[None, :, :, 3]
    return ~(sign >= 0).any(-1)


def rbbox_planes(rbbox):
    """lidar boxes [N, 7] (x, y, z, w, l, h, r) -> planes; points_in_rbbox box_np_ops.py:688-699."""
    corners = center_to_corner_box3d(rbbox[:, :3], rbbox[:, 3:6], rbbox[:, 6], origin=[0.5, 0.5, 0], axis=2)
    return planes_of_surfaces(corner_to_surfaces_3d(corners))


# ---- calibration helpers ------------------------------------------------------------------------------------
def camera_to_lidar(points, r_rect, velo2cam):
    points_shape = list(points.shape[0:-1])
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(points_shape + [1])], axis=-1)
    lidar_points = points @ np.linalg.inv((r_rect @ velo2cam).T)
    return lidar_points[..., :3]


def box_camera_to_lidar(data, r_rect, velo2cam):
    xyz = data[:, 0:3]
    l, h, w = data[:, 3:4], data[:, 4:5], data[:, 5:6]
    r = data[:, 6:7]
    return np.concatenate([camera_to_lidar(xyz, r_rect, velo2cam), w, l, h, r], axis=1)


def projection_matrix_to_CRT_kitti(proj):
    CR, CT = proj[0:3, 0:3], proj[0:3, 3]
    Rinv, Cinv = np.linalg.qr(np.linalg.inv(CR))
    return np.linalg.inv(Cinv), np.linalg.inv(Rinv), Cinv @ CT


def get_frustum(bbox_image, C, near_clip=0.001, far_clip=100):
    fku, fkv, u0v0 = C[0, 0], -C[1, 1], C[0:2, 2]
    z_points = np.array([near_clip] * 4 + [far_clip] * 4, dtype=C.dtype)[:, np.newaxis]
    b = bbox_image
    box_corners = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C.dtype)
    near = (box_corners - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=C.dtype)
    far = (box_corners - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=C.dtype)
    return np.concatenate([np.concatenate([near, far], axis=0), z_points], axis=1)


def get_frustum_v2(bboxes, C, near_clip=0.001, far_clip=100):
    fku, fkv, u0v0 = C[0, 0], -C[1, 1], C[0:2, 2]
    num_box = bboxes.shape[0]
    z_points = np.tile(np.array([near_clip] * 4 + [far_clip] * 4, dtype=C.dtype)[np.newaxis, :, np.newaxis], [num_box, 1, 1])
    box_corners = bboxes[..., [0, 1, 0, 3, 2, 3, 2, 1]].reshape(-1, 4, 2)
    near = (box_corners - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=C.dtype)
    far = (box_corners - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=C.dtype)
    return np.concatenate([np.concatenate([near, far], axis=1), z_points], axis=-1)


def image_frustum_planes(rect, Trv2c, P2, image_shape):
    """planes of remove_outside_points box_np_ops.py:629-640 (one polygon)."""
    C, R, T = projection_matrix_to_CRT_kitti(P2)
    frustum = get_frustum([0, 0, image_shape[1], image_shape[0]], C)
    frustum -= T
    frustum = np.linalg.inv(R) @ frustum.T
    frustum = camera_to_lidar(frustum.T, rect, Trv2c)
    return planes_of_surfaces(corner_to_surfaces_3d(frustum[np.newaxis, ...]))


def bbox_frustum_planes(bbox, rect, Trv2c, P2):
    """planes of get_frustum_points box_np_ops.py:643-653 (one polygon per 2D box)."""
    C, R, T = projection_matrix_to_CRT_kitti(P2)
    frustums = get_frustum_v2(bbox, C)
    frustums -= T
    frustums = np.einsum('ij, akj->aki', np.linalg.inv(R), frustums)
    frustums = camera_to_lidar(frustums, rect, Trv2c)
    return planes_of_surfaces(corner_to_surfaces_3d(frustums))


# ---- the gather itself --------------------------------------------------------------------------------------
def gather_per_box(points, planes):
    """For each polygon, in order: its inside points in original order, or ONE zero row when it has none
    (preprocess.py:76-84 / :88-94).  Returns (rows [Q, F], split [N + 1])."""
    mask = inside_planes(points, planes)
    out, split = [], [0]
    for j in range(planes.shape[0]):
        sel = points[mask[:, j]]
        if sel.shape[0] == 0:
            sel = np.zeros((1, points.shape[1]), dtype=points.dtype)
        out.append(sel)
        split.append(split[-1] + sel.shape[0])
    return np.concatenate(out, axis=0), np.asarray(split, dtype=np.int64)


def prep_points(points, rect, Trv2c, P2, img_shape, dets, use_frustum=False, without_reflectivity=False,
                det_type='3D', shift_bbox=None):
    """read_and_prep_points preprocess.py:45-106 after the velodyne file has been read into ``points``."""
    keep = inside_planes(points, image_frustum_planes(rect, Trv2c, P2, img_shape))[:, 0]
    points = points[keep]
    if det_type == '3D' and not use_frustum:
        boxes = np.concatenate([dets['location'], dets['dimensions'], dets['rotation_y'][..., np.newaxis]],
                               axis=1).astype(np.float32)
        planes = rbbox_planes(box_camera_to_lidar(boxes, rect, Trv2c))
    else:
        boxes = shift_bbox.copy() if shift_bbox is not None else dets['bbox'].copy()
        planes = bbox_frustum_planes(boxes, rect, Trv2c, P2)
    rows, split = gather_per_box(points, planes)
    if without_reflectivity:
        rows = rows[:, [0, 1, 2] + list(range(4, points.shape[1]))]
    return {'points': rows, 'points_split': split.tolist()}
