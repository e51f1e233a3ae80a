"""The LDS-DMA producer/consumer trunk kernel (conv3x3_hl16_dma.hip, all tuning variants) must agree with
the fp64 convolution like the register-staged kernel does (same arithmetic, different machine mapping)."""
import pytest
import torch

from fake_ops import TorchOps
from mmmot_amd import _lib
from mmmot_amd.pack import from_hl16, hl16_weight_shift, to_hl16
from test_kernels_gpu import close, hip, rnd  # noqa: F401  (hip is a fixture)

pytestmark = pytest.mark.gpu

CASES = [
    # pool L  H   W  Cin Cout
    (1, 2, 8, 8, 64, 64),
    (0, 2, 8, 12, 32, 128),
    (1, 3, 4, 4, 128, 256),
    (0, 5, 6, 10, 64, 64),       # 300 pixels: partial 256-row tile
    (1, 9, 4, 4, 512, 512),
    (0, 1, 16, 16, 256, 512),
]


@pytest.mark.parametrize('variant', [0, 1, 2])
@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', CASES)
def test_conv3x3_hl16_dma(hip, pool, L, H, W, Cin, Cout, variant):
    x = torch.relu(rnd(L * H * W, Cin, seed=330)) * 3.0
    w = rnd(9, Cout, Cin, seed=331, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=332, scale=0.1)
    shift = hl16_weight_shift(w)
    x16, w16 = to_hl16(x), to_hl16(w.double() * 2.0 ** shift)
    emu = TorchOps(torch.float64)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.zeros(L * Ho * Wo, Cout)
    emu.conv3x3(from_hl16(x16).view(L, H, W, Cin), (from_hl16(w16) * 2.0 ** -shift), bias, ref, L, H, W, Cin, Cout,
                False, bool(pool))
    out16 = torch.zeros(L * Ho * Wo, Cout).cuda()
    lib = _lib.load()
    lib.mmmot_set_dma_variant(variant)
    try:
        hip.conv3x3_hl16_dma(x16.cuda(), w16.cuda(), bias.cuda(), out16, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
    finally:
        lib.mmmot_set_dma_variant(0)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    close(out, ref, 2e-6, 'conv3x3 hl16 LDS-DMA kernel vs fp64')
