"""Image preparation on the GPU (csrc/crop_resize.hip through the C-ABI): bit-exact against the Pillow/torch
pipeline fixtures and against the CPU oracle on fresh frames; structural properties at full size."""
import os

import numpy as np
import pytest
import torch

from oracle import crops_ref as O
from mmmot_amd import crops as CR
from test_crops_oracle import GOLD, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_device_matches_pillow_pipeline(path):
    z, img = load(path)
    out, u8 = CR.crop_resize_normalize(torch.from_numpy(img).cuda(), z['bbox'], int(z['size']), return_u8=True)
    assert np.array_equal(u8.cpu().numpy(), z['resized_u8'])
    o = out.cpu().numpy()
    assert np.array_equal(o[0], z['out0_f32'])
    assert np.array_equal(o.astype(np.float64).sum(axis=(1, 2, 3)), z['out_sum'])


@pytest.mark.parametrize('S', [32, 64, 128, 224])
def test_device_vs_oracle_fresh_frame(S):
    rng = np.random.default_rng(S)
    H, W = 200, 330
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    xy = rng.uniform([-20, -20], [W - 10, H - 10], (9, 2))
    wh = np.concatenate([rng.uniform(3, 60, (3, 2)), rng.uniform(60, 330, (3, 2)), rng.uniform(1, 4, (3, 2))])
    bb = np.concatenate([xy, xy + wh], 1)
    u8_ref, f_ref = O.crop_resize_normalize(img, bb, S)
    out, u8 = CR.crop_resize_normalize(torch.from_numpy(img).cuda(), bb, S, return_u8=True)
    assert np.array_equal(u8.cpu().numpy(), u8_ref)
    assert np.array_equal(out.cpu().numpy(), f_ref)


def test_properties_at_full_size():
    """128 detections at 224 x 224 from a KITTI-sized frame: a constant frame resizes to the same constant
    (coefficients sum to one), a box equal to the output size is the identity crop, and the tensor equals
    (u8 / 255 - mean) / std exactly."""
    H, W, S, N = 375, 1242, 224, 128
    rng = np.random.default_rng(9)
    const = torch.full((H, W, 3), 117, dtype=torch.uint8).cuda()
    xy = rng.uniform([0, 0], [W - 260, H - 230], (N, 2))
    bb = np.concatenate([xy, xy + rng.uniform(20, 120, (N, 2))], 1)
    out, u8 = CR.crop_resize_normalize(const, bb, S, return_u8=True)
    assert int(u8.min()) == 117 and int(u8.max()) == 117
    img = torch.from_numpy(rng.integers(0, 256, (H, W, 3)).astype(np.uint8)).cuda()
    ident = np.array([[100.0, 50.0, 324.0, 274.0]])
    o2, u2 = CR.crop_resize_normalize(img, ident, S, return_u8=True)
    assert torch.equal(u2[0], img[50:274, 100:324])
    mean = torch.tensor(O.MEAN).view(1, 3, 1, 1)
    std = torch.tensor(O.STD).view(1, 3, 1, 1)
    # IEEE float32 division on the host (device torch division is not correctly rounded)
    assert torch.equal(o2.cpu(), (u2.cpu().permute(0, 3, 1, 2).float() / 255 - mean) / std)
    out3, u3 = CR.crop_resize_normalize(img, bb, S, return_u8=True)
    ref_u8, _ = O.crop_resize_normalize(img.cpu().numpy(), bb[:3], S)
    assert np.array_equal(u3[:3].cpu().numpy(), ref_u8)


@pytest.mark.parametrize('trunk', ['f16x3', 'f16q8', 'f32'])
def test_forward_from_uint8_crops_is_bitwise_the_forward_from_the_normalised_tensor(trunk):
    """SURVEY 8f rank 3 as written ("fuse into the first conv's loader"): frame + boxes -> 8-bit crops
    (mmmot_crop_resize_norm, out_u8 only) -> TrackingNet.forward; ToTensor / Normalize happen while the fused
    conv1_1 + conv1_2 launch fetches its raw window (f32 trunk: one small kernel).  Same IEEE operations as the host
    pipeline => the scores are bit-identical to feeding the fp32 `dets` tensor made by the same kernel."""
    import numpy as np
    from common import build_model, case_inputs, get_case
    from mmmot_amd.crops import crop_resize_normalize, crop_resize_u8
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, device='cuda:0')
    m.set_trunk(trunk)
    _, info, ds = case_inputs(c)
    L = sum(int(d) for d in ds)
    g = torch.Generator().manual_seed(5)
    frame = torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8).cuda()
    xy = torch.rand(L, 2, generator=g) * torch.tensor([1100.0, 300.0])
    wh = torch.rand(L, 2, generator=g) * 120 + 20
    bboxes = torch.cat([xy, xy + wh], 1).numpy().astype(np.float64)
    f32, u8ref = crop_resize_normalize(frame, bboxes, size=c['S'], return_u8=True)
    u8 = crop_resize_u8(frame, bboxes, size=c['S'])
    assert u8.dtype == torch.uint8 and torch.equal(u8, u8ref)
    dinfo = {k: v.cuda() for k, v in info.items()}
    with torch.no_grad():
        a = m(f32, dinfo, ds)
        b = m(u8, dinfo, ds)
    for x, y in ((a[0], b[0]), (a[1][0], b[1][0]), (a[2], b[2]), (a[3], b[3])):
        assert torch.equal(x, y)
