#!/usr/bin/env python
"""Generate tests/golden/crops_*.npz with Pillow + torch in this container, following the reference's call
sequence (dataset/test_seq_dataset.py:212-218, utils/build_util.py:111-112,137-142):

    transform(img.crop((floor(x1), floor(y1), ceil(x2), ceil(y2))).resize((S, S), Image.BILINEAR))
    transform = Compose([Resize(S), CenterCrop(S), ToTensor(), Normalize(mean, std)])   # first two: identities

torchvision is not installed; ToTensor / Normalize are restated with the torch calls torchvision makes
(``torch.from_numpy(np.array(pic)).permute(2, 0, 1).float().div(255)``; ``sub_(mean).div_(std)``).
The oracle (oracle/crops_ref.py) is checked bit-exact against these outputs in the same run.
"""
import os
import sys

import numpy as np
import PIL
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import crops_ref as O  # noqa: E402


def frame(seed, H, W):
    """smooth structure + fine texture, so that wrong taps / wrong rounding show up"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([127 + 90 * np.sin(x / 37.0 + c) * np.cos(y / 23.0 - c) + 30 * np.sin((x + 2 * y) / 5.0 + c)
                    for c in range(3)], -1)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def reference_pipeline(img_u8, bboxes, S):
    im = Image.fromarray(img_u8)
    mean = torch.tensor(O.MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(O.STD, dtype=torch.float32).view(3, 1, 1)
    u8s, outs = [], []
    for b in bboxes:
        x1, y1, x2, y2 = np.floor(b[0]), np.floor(b[1]), np.ceil(b[2]), np.ceil(b[3])
        pic = im.crop((x1, y1, x2, y2)).resize((S, S), Image.BILINEAR)
        arr = np.array(pic)
        t = torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)   # ToTensor
        t = t.clone().sub_(mean).div_(std)                                                   # Normalize
        u8s.append(arr)
        outs.append(t.numpy())
    return np.stack(u8s), np.stack(outs)


FRAMES = {'kitti': (21, 375, 1242), 'small': (23, 96, 128)}
CASES = [
    # name         frame    S   boxes (x1, y1, x2, y2) float
    ('crops_s64', 'kitti', 64, [[100.3, 50.7, 180.2, 259.9], [600.0, 150.0, 664.0, 214.0], [5.5, 300.2, 90.1, 374.8],
                                [-12.4, 120.0, 40.0, 200.5], [1180.7, 10.2, 1260.3, 390.0], [300.0, 100.0, 310.0, 109.0],
                                [400.2, 20.1, 1000.9, 370.3], [700.5, 170.5, 732.4, 202.6]]),
    ('crops_s224', 'kitti', 224, [[710.2, 160.9, 880.7, 301.3], [50.0, 180.0, 74.0, 199.0]]),
    ('crops_s32', 'small', 32, [[0.0, 0.0, 128.0, 96.0], [10.2, 10.9, 42.7, 44.1], [100.0, 60.0, 140.0, 110.0],
                                [3.0, 3.0, 4.0, 5.0]]),
]


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    for fname, (seed, H, W) in FRAMES.items():   # the frames are stored once, the cases refer to them by name
        np.savez_compressed(os.path.join(out_dir, 'cropframe_%s.npz' % fname), image=frame(seed, H, W))
    for name, fname, S, boxes in CASES:
        seed, H, W = FRAMES[fname]
        img = frame(seed, H, W)
        boxes = np.asarray(boxes, dtype=np.float64)
        u8, f32 = reference_pipeline(img, boxes, S)
        mine_u8, mine_f32 = O.crop_resize_normalize(img, boxes, S)
        assert np.array_equal(mine_u8, u8), name
        assert np.array_equal(mine_f32, f32), name
        # the float output is a deterministic function of the uint8 one: store the uint8 images, the first
        # float image in full and per-crop sums of the rest (fixtures stay small)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), frame=np.array(fname), bbox=boxes, size=S, resized_u8=u8,
                            out0_f32=f32[0], out_sum=f32.astype(np.float64).sum(axis=(1, 2, 3)),
                            pillow=np.array(PIL.__version__))
        print('%-11s %dx%d frame, %d boxes -> S=%d   oracle == Pillow/torch pipeline (uint8 and float32 bit-exact)' % (
            name, H, W, len(boxes), S))


if __name__ == '__main__':
    main()
