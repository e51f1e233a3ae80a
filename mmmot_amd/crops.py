"""Device image preparation: frame + 2D boxes -> the model input ``dets`` [N, 3, S, S].

Mirror of the per-detection loop of ``TestSequence._generate_img_lidar`` (reference
dataset/test_seq_dataset.py:212-218) with the valid-transform of utils/build_util.py:137-142:

    x1, y1, x2, y2 = floor(bbox[0]), floor(bbox[1]), ceil(bbox[2]), ceil(bbox[3])
    transform(img.crop((x1, y1, x2, y2)).resize((224, 224), Image.BILINEAR))

One launch pair for all detections of a frame (csrc/crop_resize.hip); the result is bit-identical to the
Pillow / torchvision pipeline and stays on the device.  No CPU fallback.
"""
import math

import numpy as np
import torch

from . import _lib
from .ops import _iptr, _ptr

MEAN = (0.485, 0.456, 0.406)   # utils/build_util.py:111-112
STD = (0.229, 0.224, 0.225)


def boxes_of_bboxes(bboxes):
    """float [N, 4] detections -> int32 [N, 4] crop boxes (floor top-left, ceil bottom-right)."""
    b = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    return np.stack([np.floor(b[:, 0]), np.floor(b[:, 1]), np.ceil(b[:, 2]), np.ceil(b[:, 3])], 1).astype(np.int32)


def crop_resize_u8(frame, bboxes, size=224):
    """The 8-bit crops only: device uint8 [N, size, size, 3] - the input ``TrackingNet.forward`` takes in place of the
    fp32 ``dets`` (ToTensor / Normalize are then applied inside the first trunk launch; the fp32 tensor never exists)."""
    return crop_resize_normalize(frame, bboxes, size, only_u8=True)


def crop_resize_normalize(frame, bboxes, size=224, mean=MEAN, std=STD, return_u8=False, only_u8=False):
    """frame: device uint8 [H, W, 3] (RGB); bboxes: [N, 4] floats (host).  Returns device fp32 [N, 3, size, size]
    (and the uint8 resized crops [N, size, size, 3] when ``return_u8``; ``only_u8``: just those)."""
    if not frame.is_cuda:
        raise RuntimeError('crop_resize_normalize needs a device frame; there is no CPU fallback')
    if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
        raise TypeError('frame must be uint8 [H, W, 3]')
    lib = _lib.load()
    frame = frame.contiguous()
    H, W = int(frame.shape[0]), int(frame.shape[1])
    boxes = boxes_of_bboxes(bboxes)
    N = boxes.shape[0]
    out = None if only_u8 else torch.empty(N, 3, size, size, dtype=torch.float32, device=frame.device)
    u8 = torch.empty(N, size, size, 3, dtype=torch.uint8, device=frame.device) if (return_u8 or only_u8) else None
    if N == 0:
        return u8 if only_u8 else ((out, u8) if return_u8 else out)
    ext = int(max((boxes[:, 2] - boxes[:, 0]).max(), (boxes[:, 3] - boxes[:, 1]).max(), 1))
    kmax = 2 * int(math.ceil(max(1.0, ext / float(size)))) + 1
    dboxes = torch.from_numpy(boxes).to(frame.device)
    ms = torch.tensor(list(mean) + list(std), dtype=torch.float32, device=frame.device)
    work = torch.empty(N * 2 * size * (2 + kmax), dtype=torch.int32, device=frame.device)
    st = lib.mmmot_crop_resize_norm(frame.data_ptr(), H, W, _iptr(dboxes), N, size, kmax, _ptr(ms), _iptr(work),
                                    None if out is None else _ptr(out), None if u8 is None else u8.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    _lib.check(st, 'mmmot_crop_resize_norm')
    return u8 if only_u8 else ((out, u8) if return_u8 else out)
