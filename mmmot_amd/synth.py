"""Seeded synthetic frame-pair inputs in the reference's input contract.

Contract (reference dataset/test_seq_dataset.py:176-246, consumed at
modules/tracking_net.py:165): ``dets`` L x 3 x S x S ImageNet-normalised crops,
frame-0 detections first; ``det_info['points']`` 1 x P x 3 raw-metre xyz;
``det_info['points_split']`` 1 x (L+1) float cumulative offsets;
``dets_split`` list of one-element int tensors [N], [M].
"""
import numpy as np
import torch


def make_pair(N, M, S, pts_per_det, seed, ragged=False, reflectivity=False):
    """One synthetic frame pair.  ``ragged`` draws per-detection point counts
    from a geometric law clipped to [1, 4*pts]; otherwise fixed.  ``reflectivity`` appends the 4th LiDAR
    channel (drawn after everything else, so the xyz / crops of a seed do not depend on it)."""
    g = np.random.Generator(np.random.Philox(key=[0x5eed, int(seed)]))
    L = N + M
    # every detection gets its own appearance (contrast, colour cast, a smooth pattern) on top of
    # pixel noise: i.i.d. noise alone makes all crops look identical to the encoder, which drives the
    # per-channel GroupNorm variances over the N x M pairs towards 0 (ill-conditioned, nothing like
    # real crops).  Range = ImageNet-normalised pixel range.
    crops = g.standard_normal((L, 3, S, S), dtype=np.float32)
    contrast = g.uniform(0.3, 1.2, (L, 1, 1, 1)).astype(np.float32)
    cast = (0.8 * g.standard_normal((L, 3, 1, 1))).astype(np.float32)
    fy, fx = g.uniform(0.5, 3.0, (2, L, 3, 1, 1)).astype(np.float32)
    ph = g.uniform(0, 2 * np.pi, (2, L, 3, 1, 1)).astype(np.float32)
    yy = np.linspace(0, 2 * np.pi, S, dtype=np.float32).reshape(1, 1, S, 1)
    xx = np.linspace(0, 2 * np.pi, S, dtype=np.float32).reshape(1, 1, 1, S)
    pattern = np.cos(fy * yy + ph[0]) * np.cos(fx * xx + ph[1])
    crops = np.clip(contrast * crops + cast + 0.9 * pattern.astype(np.float32), -2.2, 2.7).astype(np.float32)
    if ragged:
        cnt = np.clip(g.geometric(1.0 / max(pts_per_det, 1), L), 1, 4 * max(pts_per_det, 1))
        if L >= 2:
            cnt[1] = 1  # always exercise the single-point detection case
    else:
        cnt = np.full(L, pts_per_det, dtype=np.int64)
    split = np.zeros(L + 1, dtype=np.int64)
    split[1:] = np.cumsum(cnt)
    P = int(split[-1])
    lo = np.array([5.0, -20.0, -2.0], dtype=np.float32)
    hi = np.array([60.0, 20.0, 0.5], dtype=np.float32)
    centres = g.uniform(lo, hi, (L, 3)).astype(np.float32)
    sig = np.array([1.6, 0.7, 0.6], dtype=np.float32)
    pts = g.standard_normal((P, 3), dtype=np.float32) * sig
    pts += np.repeat(centres, cnt, axis=0)
    if reflectivity:  # 4th LiDAR channel in [0, 1] (without_reflectivity=False, point_cloud/preprocess.py:96-99)
        pts = np.concatenate([pts, g.uniform(0.0, 1.0, (P, 1)).astype(np.float32)], axis=1)
    dets = torch.from_numpy(crops)
    det_info = {
        'points': torch.from_numpy(pts).unsqueeze(0),
        'points_split': torch.from_numpy(split.astype(np.float32)).unsqueeze(0),
    }
    dets_split = [torch.tensor([N]), torch.tensor([M])]
    return dets, det_info, dets_split


# BASELINE.json configs -> concrete shapes (SURVEY 8 preamble).
CONFIGS = {
    'cfg1': dict(fusion='A', affinity_op='multiply', softmax_mode='none', N=10, M=12, S=224, pts=300, ragged=True, batch=1),
    'cfg2': dict(fusion='A', affinity_op='multiply', softmax_mode='none', N=32, M=32, S=64, pts=512, ragged=False, batch=32),
    'cfg3': dict(fusion='C', affinity_op='multiply', softmax_mode='none', N=64, M=64, S=128, pts=2048, ragged=False, batch=1),
    'cfg4': dict(fusion='C', affinity_op='minus_abs', softmax_mode='dual_add', N=128, M=128, S=64, pts=512, ragged=False, batch=32),
}


# ---- a synthetic KITTI-shaped SEQUENCE (camera frame + LiDAR sweep + detections per frame) ---------------------------
# KITTI tracking calibration of sequence 0000 (public dataset constants), 4x4 like the reference's info['calib/*']
KITTI_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                     [0.0, 0.0, 1.0, 0.002745884], [0.0, 0.0, 0.0, 1.0]])
KITTI_R0 = np.array([[0.9999239, 0.00983776, -0.007445048, 0.0], [-0.009869795, 0.9999421, -0.004278459, 0.0],
                     [0.007402527, 0.004351614, 0.9999631, 0.0], [0.0, 0.0, 0.0, 1.0]])
KITTI_TR = np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766],
                     [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                     [0.9998621, 0.00752379, 0.01480755, -0.2717806], [0.0, 0.0, 0.0, 1.0]])
KITTI_HW = (375, 1242)


def make_frame(seed, n_pts=120000, n_det=11, hw=KITTI_HW):
    """One frame of a synthetic sequence in the form reference dataset/test_seq_dataset.py:176-246 reads it: RGB image
    uint8 [H, W, 3], velodyne sweep fp32 [P, 4], ``frame_info`` (calibration + img_shape) and the detection dict
    (camera-frame ``location`` / ``dimensions`` / ``rotation_y`` + 2D ``bbox``).  Every 3D box holds a cluster of
    5..400 points; the 2D boxes are the projections of the box centres +- an extent, clipped loosely to the frame."""
    rng = np.random.default_rng([0x5e9, int(seed)])
    H, W = hw
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pts = np.stack([rng.uniform(0, 70, n_pts), rng.uniform(-30, 30, n_pts), rng.uniform(-2.5, 1.0, n_pts),
                    rng.uniform(0, 1, n_pts)], 1)
    loc, dims, rot, bbox, extra = [], [], [], [], []
    for _ in range(n_det):
        c = np.array([rng.uniform(6, 45), rng.uniform(-8, 8), rng.uniform(-1.9, -1.2)])  # lidar frame, bottom centre
        wlh = np.array([rng.uniform(1.4, 2.0), rng.uniform(3.2, 4.8), rng.uniform(1.3, 1.8)])
        ry = rng.uniform(-np.pi, np.pi)
        k = int(rng.integers(5, 400))
        local = rng.uniform(-0.49, 0.49, (k, 3)) * wlh
        local[:, 2] += wlh[2] / 2
        cs, sn = np.cos(ry), np.sin(ry)
        xy = local[:, :2] @ np.array([[cs, sn], [-sn, cs]])
        extra.append(np.concatenate([np.concatenate([xy, local[:, 2:3]], 1) + c, rng.uniform(0, 1, (k, 1))], 1))
        cam = (KITTI_R0 @ KITTI_TR @ np.append(c, 1.0))[:3]
        loc.append(cam)
        dims.append([wlh[1], wlh[2], wlh[0]])  # l, h, w
        rot.append(ry)
        uvw = KITTI_P2[:3] @ np.append(cam, 1.0)
        u, v = uvw[0] / uvw[2], uvw[1] / uvw[2]
        ext = 721.5 * 2.0 / max(cam[2], 1.0)
        bbox.append(np.clip([u - ext, v - ext * 0.8, u + ext, v + ext * 0.2], [-20, -20, 10, 10], [W - 40, H - 35, W + 18, H + 15]))
    pts = np.concatenate([pts] + extra, 0)
    rng.shuffle(pts)
    info = {'calib/R0_rect': KITTI_R0, 'calib/Tr_velo_to_cam': KITTI_TR, 'calib/P2': KITTI_P2, 'img_shape': np.array([H, W])}
    dets = {'location': np.asarray(loc), 'dimensions': np.asarray(dims), 'rotation_y': np.asarray(rot),
            'bbox': np.asarray(bbox)}
    return img, pts.astype(np.float32), info, dets


def make_sequence(n_frames, seed=0, n_pts=120000, det_range=(10, 12), hw=KITTI_HW):
    """``n_frames`` frames with det_range[0]..det_range[1] detections each (BASELINE cfg1's shape: N ~ 10-12)."""
    rng = np.random.default_rng([0x5ea, int(seed)])
    return [make_frame(1000 * seed + t, n_pts, int(rng.integers(det_range[0], det_range[1] + 1)), hw) for t in range(n_frames)]
