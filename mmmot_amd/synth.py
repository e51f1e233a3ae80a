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
