"""Image preparation: the CPU oracle against the Pillow/torch pipeline outputs stored in
tests/golden/crops_*.npz (oracle/gen_golden_crops.py) - no GPU needed."""
import glob
import os

import numpy as np
import pytest

from oracle import crops_ref as O

HERE = os.path.dirname(__file__)
GOLD = sorted(glob.glob(os.path.join(HERE, 'golden', 'crops_*.npz')))


def load(path):
    z = np.load(path)
    img = np.load(os.path.join(HERE, 'golden', 'cropframe_%s.npz' % str(z['frame'])))['image']
    return z, img


def test_fixtures_present():
    assert len(GOLD) >= 3


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_pillow_pipeline(path):
    z, img = load(path)
    u8, f32 = O.crop_resize_normalize(img, z['bbox'], int(z['size']))
    assert np.array_equal(u8, z['resized_u8'])            # bit-exact resized crops (incl. boxes leaving the frame)
    assert np.array_equal(f32[0], z['out0_f32'])          # bit-exact normalised tensor
    assert np.array_equal(f32.astype(np.float64).sum(axis=(1, 2, 3)), z['out_sum'])


def test_coefficients_sum_to_one_and_identity_at_scale_one():
    ks, b, kk = O.precompute_coeffs(64, 0.0, 64.0, 64)
    assert ks == 3 and (kk.sum(1) == 1 << O.PRECISION_BITS).all()
    assert (b[:, 0] == np.arange(64)).all() and (kk[:, 0] == 1 << O.PRECISION_BITS).all()
    ks, b, kk = O.precompute_coeffs(601, 0.0, 601.0, 224)      # shrinking 2.68x: support 2.68 -> 7 taps
    assert ks == 7 and np.abs(kk.sum(1) - (1 << O.PRECISION_BITS)).max() <= 3


def test_device_entry_refuses_host_tensors():
    import torch
    from mmmot_amd import crops
    with pytest.raises(RuntimeError):
        crops.crop_resize_normalize(torch.zeros(8, 8, 3, dtype=torch.uint8), [[0, 0, 4, 4]], 8)
