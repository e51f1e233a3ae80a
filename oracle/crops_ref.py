"""CPU restatement (TEST INFRASTRUCTURE ONLY - never imported by the product path) of the reference's
per-detection image preparation:

    /root/reference/dataset/test_seq_dataset.py:212-218
        x1, y1 = floor(bbox[0]), floor(bbox[1]);  x2, y2 = ceil(bbox[2]), ceil(bbox[3])
        transform(img.crop((x1, y1, x2, y2)).resize((224, 224), Image.BILINEAR))
    /root/reference/utils/build_util.py:111-112,137-142
        transform = Compose([Resize(224), CenterCrop(224), ToTensor(), Normalize(mean, std)])
        (Resize / CenterCrop are identities on a 224 x 224 image)

The arithmetic lives in two third-party packages that are not vendored in /root/reference and that the
reference does not pin (README: "torchvision", "pillow"): Pillow's ``ImagingResample`` (src/libImaging/Resample.c)
and torchvision's ``to_tensor`` / ``normalize``.  Restated here from the published algorithm:

  * ``Image.crop`` with a box that leaves the image pads with zeros;
  * ``resize(BILINEAR)`` of an 8-bit image = two separable passes (horizontal first, then vertical, an
    8-bit intermediate image in between), triangle filter whose support is stretched by the scale factor when
    shrinking (antialiasing), coefficients computed in double precision, normalised to sum 1, converted to
    22-bit fixed point with round-half-away, accumulation in int32 starting from 1 << 21, result
    clamp((acc >> 22), 0, 255);
  * ``to_tensor``: uint8 HWC -> float32 CHW / 255;  ``normalize``: (x - mean) / std in float32.

Pinned by oracle/gen_golden_crops.py against Pillow itself (the version in this container is recorded in the
fixtures; Pillow's resampling code has been stable since 7.x) - bit-exact on every fixture.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _bilinear(x):
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size, in0, in1, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1.0).
    Returns (ksize, bounds [out_size][2] = (xmin, count), kk int32 [out_size][ksize])."""
    scale = float(in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def crop_u8(img, box):
    """PIL Image.crop: img uint8 [H][W][3], box (x1, y1, x2, y2) ints -> [y2-y1][x2-x1][3], zeros outside."""
    x1, y1, x2, y2 = box
    H, W = img.shape[:2]
    out = np.zeros((max(y2 - y1, 0), max(x2 - x1, 0), img.shape[2]), dtype=np.uint8)
    sx0, sy0, sx1, sy1 = max(x1, 0), max(y1, 0), min(x2, W), min(y2, H)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y1:sy1 - y1, sx0 - x1:sx1 - x1] = img[sy0:sy1, sx0:sx1]
    return out


def _pass(src, bounds, kk, axis):
    """one separable pass over ``axis`` (1: horizontal, 0: vertical) of a uint8 image, vectorised per output index"""
    n_out = bounds.shape[0]
    shape = list(src.shape)
    shape[axis] = n_out
    out = np.zeros(shape, dtype=np.uint8)
    s = src.astype(np.int64)
    for o in range(n_out):
        x0, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(np.take(s, 0, axis=axis).shape, 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i in range(cnt):
            acc = acc + np.take(s, x0 + i, axis=axis) * int(kk[o, i])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o] = v
        else:
            out[o] = v
    return out


def resize_bilinear_u8(crop, size):
    """PIL ``resize((size, size), Image.BILINEAR)`` of a uint8 [h][w][3] image."""
    h, w = crop.shape[:2]
    _, bx, kx = precompute_coeffs(w, 0.0, float(w), size)
    _, by, ky = precompute_coeffs(h, 0.0, float(h), size)
    tmp = _pass(crop, bx, kx, axis=1)      # horizontal first (Resample.c ImagingResampleInner)
    return _pass(tmp, by, ky, axis=0)


def box_of_bbox(bbox):
    """test_seq_dataset.py:212-215: floor the top-left, ceil the bottom-right."""
    return (int(np.floor(bbox[0])), int(np.floor(bbox[1])), int(np.ceil(bbox[2])), int(np.ceil(bbox[3])))


def to_tensor_normalize(u8):
    """torchvision to_tensor + normalize on a uint8 [S][S][3] image -> float32 [3][S][S]."""
    x = u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    mean = np.asarray(MEAN, dtype=np.float32)[:, None, None]
    std = np.asarray(STD, dtype=np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def crop_resize_normalize(img, bboxes, size=224):
    """img uint8 [H][W][3]; bboxes float [N][4] -> (uint8 [N][size][size][3], float32 [N][3][size][size])."""
    u8 = np.stack([resize_bilinear_u8(crop_u8(img, box_of_bbox(b)), size) for b in bboxes])
    return u8, np.stack([to_tensor_normalize(u) for u in u8])
