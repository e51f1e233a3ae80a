"""Device point-cloud gather: the step of the reference tracker that turns one LiDAR sweep plus the frame's
detections into ``det_info['points']`` / ``det_info['points_split']``.

Mirror of ``read_and_prep_points`` (reference point_cloud/preprocess.py:45-106) minus the velodyne file read:
same arguments, same result dict, but the O(points x boxes x 6) membership test and the ordered per-box
compaction run on the GPU (mmmot_points_count / mmmot_points_scatter, csrc/points_gather.hip) and the result
stays on the device, ready for ``TrackingNet.forward``.  What remains on the host is O(boxes): calibration
algebra and the plane equations of every box / frustum (float64 numpy, the operations of
point_cloud/box_np_ops.py:147-178,236-254,312-337,442-493,584-618,702-720 and geometry.py:84-93 in the same
order so that the planes - and therefore every inside/outside decision - agree with the reference).
"""
import threading

import numpy as np
import torch

from . import _lib
from .ops import _iptr, _ptr

_SURF_IDX = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)
MAX_POLY = 256


# ---- host geometry: boxes / frustums -> inward-facing plane equations --------------------------------------
def _planes(corners):
    """corners [N, 8, 3] -> float64 [N, 6, 4] rows (nx, ny, nz, d)."""
    surf = corners[:, _SURF_IDX][:, :, :3, :]
    vec = surf[:, :, :2, :] - surf[:, :, 1:3, :]
    normal = np.cross(vec[:, :, 0, :], vec[:, :, 1, :])
    d = -np.einsum('aij, aij->ai', normal, surf[:, :, 0, :])
    return np.ascontiguousarray(np.concatenate([normal, d[..., None]], axis=-1), dtype=np.float64)


def _cam_to_lidar(xyz, rect, Trv2c):
    if xyz.shape[-1] == 3:
        xyz = np.concatenate([xyz, np.ones(list(xyz.shape[:-1]) + [1])], axis=-1)  # float64 from here on
    return (xyz @ np.linalg.inv((rect @ Trv2c).T))[..., :3]


def _crt(P2):
    CR, CT = P2[0:3, 0:3], P2[0:3, 3]
    Rinv, Cinv = np.linalg.qr(np.linalg.inv(CR))
    return np.linalg.inv(Cinv), np.linalg.inv(Rinv), Cinv @ CT


def _frustum_xyz(corners_uv, C, near=0.001, far=100):
    """pixel corners [..., 4, 2] -> camera-frame frustum corners [..., 8, 3] (4 near, 4 far)."""
    fku, fkv, u0v0 = C[0, 0], -C[1, 1], C[0:2, 2]
    n = (corners_uv - u0v0) / np.array([fku / near, -fkv / near], dtype=C.dtype)
    f = (corners_uv - u0v0) / np.array([fku / far, -fkv / far], dtype=C.dtype)
    xy = np.concatenate([n, f], axis=-2)
    z = np.broadcast_to(np.array([near] * 4 + [far] * 4, dtype=C.dtype)[:, None], xy.shape[:-1] + (1,))
    return np.concatenate([xy, z], axis=-1)


def rbbox_planes(boxes_cam, rect, Trv2c):
    """camera-frame KITTI boxes [N, 7] (x, y, z, l, h, w, ry), float32 -> planes of the lidar-frame boxes."""
    xyz = _cam_to_lidar(boxes_cam[:, 0:3], rect, Trv2c)
    l, h, w, r = boxes_cam[:, 3:4], boxes_cam[:, 4:5], boxes_cam[:, 5:6], boxes_cam[:, 6]
    lidar = np.concatenate([xyz, w, l, h, boxes_cam[:, 6:7]], axis=1)
    dims, ang = lidar[:, 3:6], lidar[:, 6]
    unit = np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1).astype(dims.dtype)[[0, 1, 3, 2, 4, 5, 7, 6]]
    corners = dims.reshape(-1, 1, 3) * (unit - np.array([0.5, 0.5, 0], dtype=dims.dtype)).reshape(1, 8, 3)
    s, c = np.sin(ang), np.cos(ang)
    one, zero = np.ones_like(c), np.zeros_like(c)
    rot_T = np.stack([[c, -s, zero], [s, c, zero], [zero, zero, one]])
    corners = np.einsum('aij,jka->aik', corners, rot_T)
    corners += lidar[:, :3].reshape(-1, 1, 3)
    return _planes(corners)


def image_frustum_planes(rect, Trv2c, P2, image_shape):
    """The camera's viewing frustum (remove_outside_points, box_np_ops.py:629-640): one polygon."""
    C, R, T = _crt(P2)
    b = [0, 0, image_shape[1], image_shape[0]]
    uv = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C.dtype)
    fr = _frustum_xyz(uv, C)
    fr = fr - T
    fr = np.linalg.inv(R) @ fr.T
    return _planes(_cam_to_lidar(fr.T, rect, Trv2c)[np.newaxis, ...])


def bbox_frustum_planes(bbox, rect, Trv2c, P2):
    """Frustums of 2D boxes [N, 4] (get_frustum_points, box_np_ops.py:643-653)."""
    C, R, T = _crt(P2)
    uv = bbox[..., [0, 1, 0, 3, 2, 3, 2, 1]].reshape(-1, 4, 2)
    fr = _frustum_xyz(uv, C)
    fr = fr - T
    fr = np.einsum('ij, akj->aki', np.linalg.inv(R), fr)
    return _planes(_cam_to_lidar(fr, rect, Trv2c))


# ---- device gather ------------------------------------------------------------------------------------------
def gather_points(points, planes, pad_empty=True, drop_reflectivity=False):
    """points: device fp32 [P, F] (F = 3 or 4); planes: numpy/torch float64 [N, 6, 4].
    Returns (rows [Q, Fo] device fp32, split: numpy int64 [N + 1]).  No CPU fallback."""
    if not points.is_cuda:
        raise RuntimeError('gather_points needs a device tensor; there is no CPU fallback')
    lib = _lib.load()
    points = points.contiguous()
    P, F = int(points.shape[0]), int(points.shape[1])
    Fo = 3 if (drop_reflectivity and F == 4) else F
    pl = torch.as_tensor(np.ascontiguousarray(planes, dtype=np.float64)).to(points.device)
    N = int(pl.shape[0])
    if P == 0:  # nothing to test: every polygon is empty
        n = N if pad_empty else 0
        return torch.zeros(n, Fo, device=points.device), np.arange(N + 1, dtype=np.int64) * (1 if pad_empty else 0)
    stream = torch.cuda.current_stream().cuda_stream
    nblk = (P + 255) // 256
    rows, splits, base = [], [0], 0
    for j0 in range(0, N, MAX_POLY):  # the kernel holds at most 256 polygons in LDS
        nb = min(MAX_POLY, N - j0)
        cnt = torch.empty(nb * nblk + nb, dtype=torch.int32, device=points.device)
        split = torch.empty(nb + 1, dtype=torch.int32, device=points.device)
        chunk = pl[j0:j0 + nb].contiguous()
        _lib.check(lib.mmmot_points_count(_ptr(points), P, F, chunk.data_ptr(), nb, int(pad_empty), _iptr(cnt),
                                          _iptr(split), stream), 'mmmot_points_count')
        h_split = split.cpu().numpy().astype(np.int64)  # the one D2H: the host plan needs points_split anyway
        out = torch.empty(int(h_split[-1]), Fo, dtype=torch.float32, device=points.device)
        if h_split[-1] > 0:
            _lib.check(lib.mmmot_points_scatter(_ptr(points), P, F, chunk.data_ptr(), nb, _iptr(cnt), _iptr(split),
                                                _ptr(out), Fo, stream), 'mmmot_points_scatter')
        rows.append(out)
        splits.extend((base + h_split[1:]).tolist())
        base += int(h_split[-1])
    return (rows[0] if len(rows) == 1 else torch.cat(rows)), np.asarray(splits, dtype=np.int64)


def prep_points(points, info, dets, use_frustum=False, without_reflectivity=False, det_type='3D', shift_bbox=None):
    """``read_and_prep_points`` (preprocess.py:45-106) for a sweep that is already in device memory.

    points: device fp32 [P, F]; info: the reference's frame_info dict ('calib/R0_rect', 'calib/Tr_velo_to_cam',
    'calib/P2', 'img_shape'); dets: 'location', 'dimensions', 'rotation_y' (3D boxes) or 'bbox'.
    Returns {'points': device [Q, 3|4], 'points_split': list of N + 1 ints} like the reference."""
    rect = np.asarray(info['calib/R0_rect']).astype(np.float32)
    Trv2c = np.asarray(info['calib/Tr_velo_to_cam']).astype(np.float32)
    P2 = np.asarray(info['calib/P2']).astype(np.float32)
    kept, _ = gather_points(points, image_frustum_planes(rect, Trv2c, P2, info['img_shape']), pad_empty=False)
    if det_type == '3D' and not use_frustum:
        boxes = np.concatenate([dets['location'], dets['dimensions'], np.asarray(dets['rotation_y'])[..., np.newaxis]],
                               axis=1).astype(np.float32)
        planes = rbbox_planes(boxes, rect, Trv2c)
    else:
        boxes = np.asarray(shift_bbox if shift_bbox is not None else dets['bbox']).copy()
        planes = bbox_frustum_planes(boxes, rect, Trv2c, P2)
    rows, split = gather_points(kept, planes, pad_empty=True, drop_reflectivity=without_reflectivity)
    return {'points': rows, 'points_split': split.tolist()}


# ---- batched: many sweeps per launch, image-frustum filter fused into the per-box test -----------------------
_STAGE = threading.local()


def _staging(nbytes):
    """Pinned host buffer for the per-batch table upload, grown geometrically, ONE PER THREAD: the buffer is filled on
    the host and copied with non_blocking=True, and what makes its reuse safe is the ``split.cpu()`` of the same call,
    which synchronises the CALLING thread's stream behind that copy.  A process-wide buffer (round 4) could be
    overwritten - or replaced on growth - by a second thread while the first one's copy was still queued behind trunk
    work (the library supports one model per thread: ``HipOps._tls``)."""
    t = getattr(_STAGE, 'buf', None)
    if t is None or t.numel() < nbytes:
        t = _STAGE.buf = torch.empty(max(1 << 16, 2 * nbytes), dtype=torch.uint8).pin_memory()
    return t


def gather_points_batched(points, sweep_rows, planes, poly_counts, filters=None, pad_empty=True,
                          drop_reflectivity=False):
    """points: device fp32 [sum P_s, F] (the sweeps concatenated); sweep_rows: NS + 1 row offsets;
    planes: float64 [NPOLY(+filters), 6, 4]; poly_counts: polygons per sweep (they are consecutive in ``planes``,
    at most 256 per sweep); filters: per sweep the index in ``planes`` of a polygon that must also contain every
    emitted point, or -1.  Returns (rows [Q, Fo] device, split int64 [NPOLY + 1]) with the polygons of all sweeps
    in order - ONE split read-back for the whole batch."""
    if not points.is_cuda:
        raise RuntimeError('gather_points_batched needs a device tensor; there is no CPU fallback')
    lib = _lib.load()
    points = points.contiguous()
    F = int(points.shape[1])
    Fo = 3 if (drop_reflectivity and F == 4) else F
    sweep_rows = np.asarray(sweep_rows, dtype=np.int64)
    NS = len(sweep_rows) - 1
    poly_counts = np.asarray(poly_counts, dtype=np.int64)
    if len(poly_counts) != NS or (poly_counts > MAX_POLY).any() or (poly_counts < 1).any():
        raise ValueError('every sweep needs 1..%d polygons' % MAX_POLY)
    poly0 = np.concatenate([[0], np.cumsum(poly_counts)])
    NPOLY = int(poly0[-1])
    filt = np.full(NS, -1, dtype=np.int64) if filters is None else np.asarray(filters, dtype=np.int64)
    nblk = (np.diff(sweep_rows) + 255) // 256
    if (nblk < 1).any():
        raise ValueError('every sweep needs at least one point')
    blk_first = np.concatenate([[0], np.cumsum(nblk)])
    NBLK = int(blk_first[-1])
    blk_sweep = np.repeat(np.arange(NS), nblk)
    cnt_off = np.concatenate([[0], np.cumsum(np.repeat(nblk, poly_counts))])
    cnt_total = int(cnt_off[-1])
    dev = points.device
    # ONE upload per batch: the six integer tables and the float64 plane table travel in one pinned staging buffer and
    # one asynchronous copy
    tabs = [np.ascontiguousarray(t, dtype=np.int32) for t in (blk_sweep, blk_first[:-1], sweep_rows, poly0, filt, cnt_off[:-1])]
    pl_h = np.ascontiguousarray(planes, dtype=np.float64)
    offs, o = [], 0
    for t in tabs:
        offs.append(o)
        o += (t.nbytes + 15) // 16 * 16
    pl_off = o
    total = pl_off + pl_h.nbytes
    stage = _staging(total)
    hb = stage.numpy()
    for t, off in zip(tabs, offs):
        hb[off:off + t.nbytes] = t.view(np.uint8).reshape(-1)
    hb[pl_off:total] = pl_h.view(np.uint8).reshape(-1)
    dbuf = torch.empty(total, dtype=torch.uint8, device=dev)
    dbuf.copy_(stage[:total], non_blocking=True)
    d_blk_sweep, d_blk_first, d_rows, d_poly0, d_filt, d_off = [
        dbuf[off:off + t.nbytes].view(torch.int32) for t, off in zip(tabs, offs)]
    pl = dbuf[pl_off:total].view(torch.float64)
    # counters, totals and (8-byte aligned) the four 64-bit wave masks per counter the scatter pass reads back
    cnt = torch.empty(((cnt_total + NPOLY + 1) & ~1) + 8 * cnt_total, dtype=torch.int32, device=dev)
    split = torch.empty(NPOLY + 1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    common = (_ptr(points), F, NS, NPOLY, NBLK, cnt_total, pl.data_ptr(), _iptr(d_blk_sweep), _iptr(d_blk_first),
              _iptr(d_rows), _iptr(d_poly0), _iptr(d_filt), _iptr(d_off))
    _lib.check(lib.mmmot_points_count_batched(*common, int(pad_empty), _iptr(cnt), _iptr(split), stream),
               'mmmot_points_count_batched')
    h_split = split.cpu().numpy().astype(np.int64)  # the one D2H of the batch
    out = torch.empty(int(h_split[-1]), Fo, dtype=torch.float32, device=dev)
    if h_split[-1] > 0:
        _lib.check(lib.mmmot_points_scatter_batched(*common, _iptr(cnt), _iptr(split), _ptr(out), Fo, stream),
                   'mmmot_points_scatter_batched')
    return out, h_split


def prep_points_batched(sweeps, infos, dets_list, use_frustum=False, without_reflectivity=False, det_type='3D',
                        shift_bboxes=None):
    """``read_and_prep_points`` for a batch of frames in one launch sequence: sweeps = list of device [P_s, F]
    tensors (or one concatenated tensor plus row offsets as a tuple).  Returns one reference-style dict per frame
    ({'points': device rows, 'points_split': list}); the rows of all frames live in one device buffer."""
    if isinstance(sweeps, tuple):
        points, rows = sweeps
    else:
        rows = np.concatenate([[0], np.cumsum([int(s.shape[0]) for s in sweeps])])
        points = torch.cat(list(sweeps))
    planes, counts, filt = [], [], []
    base = 0
    for i, (info, dets) in enumerate(zip(infos, dets_list)):
        rect = np.asarray(info['calib/R0_rect']).astype(np.float32)
        Trv2c = np.asarray(info['calib/Tr_velo_to_cam']).astype(np.float32)
        P2 = np.asarray(info['calib/P2']).astype(np.float32)
        if det_type == '3D' and not use_frustum:
            boxes = np.concatenate([dets['location'], dets['dimensions'],
                                    np.asarray(dets['rotation_y'])[..., np.newaxis]], axis=1).astype(np.float32)
            pl = rbbox_planes(boxes, rect, Trv2c)
        else:
            sb = None if shift_bboxes is None else shift_bboxes[i]
            pl = bbox_frustum_planes(np.asarray(sb if sb is not None else dets['bbox']).copy(), rect, Trv2c, P2)
        planes.append(pl)
        counts.append(pl.shape[0])
        base += pl.shape[0]
    # the image frustums follow the boxes in the plane table
    for i, info in enumerate(infos):
        rect = np.asarray(info['calib/R0_rect']).astype(np.float32)
        Trv2c = np.asarray(info['calib/Tr_velo_to_cam']).astype(np.float32)
        P2 = np.asarray(info['calib/P2']).astype(np.float32)
        planes.append(image_frustum_planes(rect, Trv2c, P2, info['img_shape']))
        filt.append(base + i)
    out, split = gather_points_batched(points, rows, np.concatenate(planes), counts, filters=filt, pad_empty=True,
                                       drop_reflectivity=without_reflectivity)
    res, p0 = [], 0
    for c in counts:
        lo, hi = int(split[p0]), int(split[p0 + c])
        res.append({'points': out[lo:hi], 'points_split': (split[p0:p0 + c + 1] - lo).tolist()})
        p0 += c
    return res
