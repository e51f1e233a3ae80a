"""The LDS-resident-patch trunk kernel (conv3x3_hl16_patch.hip) must agree with the fp64 convolution.  Geometry cases exercise the three block
shapes (16x16x1, 8x8x4, whole maps of at most 4x4 x16), partial blocks (maps that are not multiples of the block), maps smaller than a block,
tiles that straddle crops, partial last tiles and 2..16 channel slabs."""
import pytest
import torch

from fake_ops import TorchOps
from mmmot_amd.pack import conv1_weight_shift, from_hl16, hl16_weight_shift, to_hl16  # noqa: F401
from test_kernels_gpu import close, hip, rnd  # noqa: F401  (hip is a fixture)

pytestmark = pytest.mark.gpu

CASES = [
    # pool L  H   W  Cin Cout
    (1, 2, 8, 8, 64, 64),        # 8x8 blocks, half-filled tile (2 of 4 blocks)
    (0, 5, 8, 8, 32, 128),       # 8x8 blocks, tile straddles crops, partial last tile, single slab
    (1, 3, 4, 4, 128, 256),      # map smaller than a block
    (0, 3, 2, 2, 64, 64),        # 2x2 maps (S=32 crops at conv5)
    (0, 5, 6, 10, 64, 64),       # 16x16 blocks, partial block
    (1, 2, 14, 14, 64, 128),     # 14x14 (S=224 crops at conv5)
    (0, 1, 16, 16, 256, 512),    # exactly one block per crop, 4 channel tiles
    (1, 2, 32, 32, 128, 128),    # 2x2 blocks per crop
    (0, 2, 28, 20, 64, 64),      # partial blocks on both axes
    (1, 9, 4, 4, 512, 512),      # 16 slabs
    (1, 1, 64, 64, 64, 64),      # conv1_2-like
    # odd maps (crop sides that are not a multiple of 32): pooled layers floor like nn.MaxPool2d(2, 2)
    (0, 3, 25, 25, 64, 64),      # 100-pixel crops at conv3 (16x16 blocks, partial, odd)
    (1, 3, 25, 25, 128, 128),    # ... pooled: 25 -> 12, the last row / column has no window
    (1, 4, 5, 5, 64, 128),       # 40-pixel crops at conv4 (8x8 blocks): 5 -> 2
    (0, 2, 3, 7, 32, 64),        # odd and different sides
    (1, 2, 9, 17, 64, 64),       # both block shapes' borders: 9 -> 4, 17 -> 8
]


def run_case(hip, pool, L, H, W, Cin, Cout, seed=430):
    x = torch.relu(rnd(L * H * W, Cin, seed=seed)) * 3.0
    w = rnd(9, Cout, Cin, seed=seed + 1, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=seed + 2, scale=0.1)
    shift = hl16_weight_shift(w)
    x16, w16 = to_hl16(x), to_hl16(w.double() * 2.0 ** shift)
    emu = TorchOps(torch.float64)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.zeros(L * Ho * Wo, Cout)
    emu.conv3x3(from_hl16(x16).view(L, H, W, Cin), (from_hl16(w16) * 2.0 ** -shift), bias, ref, L, H, W, Cin, Cout,
                False, bool(pool))
    out16 = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), out16, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    return out, ref, (x16, w16, bias, shift)


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', CASES)
def test_conv3x3_hl16_patch(hip, pool, L, H, W, Cin, Cout):
    out, ref, _ = run_case(hip, pool, L, H, W, Cin, Cout)
    close(out, ref, 2e-6, 'conv3x3 hl16 patch kernel vs fp64')


# more tiles than workgroups: the tiles of a workgroup are chained (the successor's first patch slab and weight
# stages are streamed in during the last slab) - forced here by capping the persistent grid
CHAIN_CASES = CASES + [
    (0, 7, 16, 16, 32, 128),     # single-slab tiles: every slab is a transition slab
    (1, 12, 32, 32, 96, 64),     # odd slab count (the two patch buffers swap roles from tile to tile), 64-channel tiles
    (0, 40, 16, 16, 128, 256),   # 80 items on 8 workgroups: 10-tile chains, channel tile changes inside a chain
    (1, 33, 24, 40, 64, 128),    # partial blocks + ragged last tile inside chains
]


@pytest.fixture
def small_grid(hip):
    from mmmot_amd import _lib
    lib = _lib.load()
    assert lib.mmmot_set_patch_grid_limit(8) == 0
    yield 8
    assert lib.mmmot_set_patch_grid_limit(0) == 0


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', CHAIN_CASES)
def test_conv3x3_hl16_patch_chained_tiles(hip, small_grid, pool, L, H, W, Cin, Cout):
    out, ref, (x16, w16, bias, shift) = run_case(hip, pool, L, H, W, Cin, Cout, seed=470)
    close(out, ref, 2e-6, 'chained tiles (grid capped at 8 workgroups) vs fp64')
    # and bitwise the same as the unchained launch (one workgroup per tile where the grid allows)
    from mmmot_amd import _lib
    _lib.load().mmmot_set_patch_grid_limit(0)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    o2 = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), o2, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
    u2 = torch.zeros_like(o2)
    hip.hl16_unpack(o2, u2)
    assert torch.equal(out.cpu(), u2.cpu()), 'chained and unchained launches differ'


# maps of at most 4 x 4 pixels (conv5 at 64-pixel crops, conv4 / conv5 at 32-pixel crops): the whole-map geometry -
# 16 maps per 256-row tile, no halo in LDS, taps that leave the map read a zero page
WHOLE_CASES = [
    # pool L  H  W  Cin Cout
    (0, 16, 4, 4, 64, 128),      # exactly one full tile
    (1, 16, 4, 4, 64, 128),
    (0, 40, 4, 4, 64, 128),      # 2.5 tiles: ragged last tile
    (1, 37, 4, 4, 96, 64),       # odd slab count, 64-channel tiles
    (0, 1, 4, 4, 32, 64),        # one map, single slab
    (0, 5, 3, 3, 64, 64),        # maps smaller than the block: unused pixel slots stay zero
    (1, 6, 4, 2, 64, 128),       # different sides
    (1, 7, 3, 4, 64, 64),        # odd side, pooled: 3 -> 1
    (0, 19, 1, 1, 64, 64),       # 1 x 1 maps (conv5_x output side at 32-pixel crops is 2, pooled to 1)
    (0, 18, 2, 2, 128, 128),     # 2 x 2 maps (conv5 at 32-pixel crops)
    (1, 18, 2, 2, 128, 128),
    (0, 70, 4, 4, 512, 512),     # conv5_1 / conv5_2 shape, several tiles x 4 channel tiles
    (1, 70, 4, 4, 512, 512),     # conv5_3
]


def _launch(hip, x16, w16, bias, shift, pool, L, H, W, Cin, Cout):
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    o = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), o, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
    u = torch.zeros_like(o)
    hip.hl16_unpack(o, u)
    return u.cpu()


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', WHOLE_CASES)
def test_conv3x3_hl16_patch_whole_map_blocks(hip, pool, L, H, W, Cin, Cout):
    from mmmot_amd import _lib
    lib = _lib.load()
    out, ref, (x16, w16, bias, shift) = run_case(hip, pool, L, H, W, Cin, Cout, seed=490)
    close(out, ref, 2e-6, 'whole-map 4x4 blocks vs fp64')
    # bit for bit the result of the haloed 8 x 8 geometry (exact zeros from the halo, same accumulation order) ...
    assert lib.mmmot_set_patch_min_block(8) == 0
    try:
        o8 = _launch(hip, x16, w16, bias, shift, pool, L, H, W, Cin, Cout)
    finally:
        assert lib.mmmot_set_patch_min_block(0) == 0
    assert torch.equal(out.cpu(), o8), 'whole-map and haloed 8x8 geometries differ'
    # ... and of the chained launch (grid capped at 8 workgroups: the successor's patch slab streams in during the last slab)
    assert lib.mmmot_set_patch_grid_limit(8) == 0
    try:
        oc = _launch(hip, x16, w16, bias, shift, pool, L, H, W, Cin, Cout)
    finally:
        assert lib.mmmot_set_patch_grid_limit(0) == 0
    assert torch.equal(out.cpu(), oc), 'chained and unchained whole-map launches differ'


def test_whole_map_blocks_long_chains_are_deterministic(hip):
    """600 maps of 4 x 4 on 8 workgroups: 38 pixel tiles x 4 channel tiles = 19-tile chains per workgroup, repeated."""
    from mmmot_amd import _lib
    lib = _lib.load()
    pool, L, H, W, Cin, Cout = 0, 600, 4, 4, 128, 512
    out, ref, (x16, w16, bias, shift) = run_case(hip, pool, L, H, W, Cin, Cout, seed=495)
    close(out, ref, 2e-6, 'whole-map blocks, one workgroup per tile')
    assert lib.mmmot_set_patch_grid_limit(8) == 0
    try:
        for _ in range(5):
            oc = _launch(hip, x16, w16, bias, shift, pool, L, H, W, Cin, Cout)
            assert torch.equal(out.cpu(), oc), 'chained whole-map launch differs'
    finally:
        assert lib.mmmot_set_patch_grid_limit(0) == 0


def test_patch_kernel_is_deterministic(hip):
    """repeated launches of the patch kernel are bitwise identical (no race in the DMA ring / counted-vmcnt
    pipeline)."""
    pool, L, H, W, Cin, Cout = 0, 6, 16, 16, 256, 256
    out, ref, (x16, w16, bias, shift) = run_case(hip, pool, L, H, W, Cin, Cout, seed=440)
    xs, ws, bs = x16.cuda(), w16.cuda(), bias.cuda()
    first = None
    for _ in range(20):
        o = torch.zeros(L * H * W, Cout).cuda()
        hip.conv3x3_hl16_patch(xs, ws, bs, o, L, H, W, Cin, Cout, False, 2.0 ** -shift)
        if first is None:
            first = o.clone()
        else:
            assert torch.equal(first, o), 'patch kernel is not deterministic across launches'


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', [(0, 3, 16, 16, 64, 128), (1, 2, 8, 8, 128, 64), (1, 2, 32, 32, 64, 256)])
def test_conv3x3_hl16_patch_per_channel_scales(hip, pool, L, H, W, Cin, Cout):
    """Output channels with gains spread over 1e6 (a trained, BatchNorm-folded layer): every channel carries its own
    power-of-two weight scale (pack.hl16_channel_shifts) and the [Cout] vector undoes it in the epilogue; each channel
    must keep fp32-class RELATIVE accuracy."""
    from mmmot_amd.pack import hl16_channel_shifts
    g = torch.Generator().manual_seed(77)
    gain = torch.pow(10.0, torch.rand(Cout, generator=g) * 6.0 - 4.0).double()     # 1e-4 .. 1e2
    x = torch.relu(rnd(L * H * W, Cin, seed=480)) * 3.0
    w = rnd(9, Cout, Cin, seed=481, scale=(2.0 / (9 * Cin)) ** 0.5).double() * gain.view(1, -1, 1)
    bias = (rnd(Cout, seed=482, scale=0.1).double() * gain).float()
    shifts = hl16_channel_shifts(w)
    assert int(shifts.max() - shifts.min()) >= 15
    w16 = to_hl16(w * torch.pow(2.0, shifts.double()).view(1, -1, 1))
    osc = torch.pow(2.0, -shifts.double()).float()
    x16 = to_hl16(x)
    emu = TorchOps(torch.float64)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.zeros(L * Ho * Wo, Cout, dtype=torch.float64)
    wv = from_hl16(w16).double() * osc.double().view(1, -1, 1)
    emu.conv3x3(from_hl16(x16).view(L, H, W, Cin), wv, bias, ref, L, H, W, Cin, Cout, False, bool(pool))
    out16 = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), out16, L, H, W, Cin, Cout, bool(pool), osc.cuda())
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    # per channel: fp32-class relative to the channel's largest output, plus the absolute quantum of the hl16 OUTPUT
    # format (the lo half of a value below ~2.5e-4 is an fp16 subnormal: 2^-25 absolute, see
    # test_hl16_small_magnitudes_keep_absolute_accuracy)
    err, chmax = (out.cpu().double() - ref).abs().amax(dim=0), ref.abs().amax(dim=0)
    rel = (err / chmax.clamp_min(1e-30)).max().item()
    print('per-channel-scale conv: worst per-channel relative error %.2e (output maxima %.1e .. %.1e)' % (
        rel, chmax.min().item(), chmax.max().item()))
    assert (err <= 4e-6 * chmax + 2.0 ** -24).all(), rel


@pytest.mark.parametrize('L,H,W', [(2, 16, 16), (3, 32, 48), (1, 14, 22), (5, 64, 64), (2, 8, 8)])
def test_conv1_fused_matches_two_layer_reference(hip, L, H, W):
    """conv1_1 (3->64) + conv1_2 (64->64) + max-pool in one launch vs the float64 two-layer computation; maps that
    are not multiples of the 16x16 block, crops smaller than a block, tiles at image borders."""
    crops = rnd(L, 3, H, W, seed=500) * 1.5
    w1 = rnd(64, 3, 3, 3, seed=501, scale=(2.0 / 27) ** 0.5)
    b1 = rnd(64, seed=502, scale=0.1)
    w2 = rnd(9, 64, 64, seed=503, scale=(2.0 / 576) ** 0.5)
    b2 = rnd(64, seed=504, scale=0.1)
    w1p = torch.zeros(64, 32)
    w1p[:, :27] = w1.permute(0, 2, 3, 1).reshape(64, 27)
    s1, s2 = conv1_weight_shift(w1p, b1), hl16_weight_shift(w2)
    w1h, w2h = to_hl16(w1p.double() * 2.0 ** s1), to_hl16(w2.double() * 2.0 ** s2)
    # float64 reference built from the SAME (hl16-rounded) weights
    w1r = (from_hl16(w1h) * 2.0 ** -s1)[:, :27].view(64, 3, 3, 3).permute(0, 3, 1, 2).double()
    w2r = (from_hl16(w2h.reshape(9 * 64, 64)) * 2.0 ** -s2).view(3, 3, 64, 64).permute(2, 3, 0, 1).double()
    y = torch.relu(torch.nn.functional.conv2d(crops.double(), w1r, b1.double(), padding=1))
    y = torch.relu(torch.nn.functional.conv2d(y, w2r, b2.double(), padding=1))
    y = torch.nn.functional.max_pool2d(y, 2, 2)
    ref = y.permute(0, 2, 3, 1).reshape(-1, 64).float()
    out16 = torch.full((L * (H // 2) * (W // 2), 64), float('nan')).cuda()
    hip.conv1_fused_hl16(crops.cuda(), w1h.cuda(), b1.cuda(), 2.0 ** -s1, w2h.cuda(), b2.cuda(), 2.0 ** -s2, out16, L, H, W)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    close(out, ref, 3e-6, 'fused conv1_1 + conv1_2 + pool vs float64')


def test_conv1_fused_many_tiles_per_workgroup(hip, small_grid):
    """persistent grid capped at 8 workgroups: every workgroup walks 10+ tiles (raw-window prefetch of the next tile)"""
    test_conv1_fused_matches_two_layer_reference(hip, 5, 64, 64)
    test_conv1_fused_matches_two_layer_reference(hip, 3, 32, 48)


def test_tile_width_heuristic_is_bitwise_neutral(hip):
    """Small problems run 64-channel tiles (twice as many items for the 256 CUs), large ones 128-channel tiles: the
    launcher decides from the problem size alone (pt_use_bn64) and the per-element arithmetic is the same - the first
    two crops of a 512-crop launch (128-channel tiles) equal a 2-crop launch (64-channel tiles) bit for bit."""
    H = W = 16
    Cin, Cout = 64, 128
    x = torch.relu(rnd(512 * H * W, Cin, seed=600)) * 3.0
    w = rnd(9, Cout, Cin, seed=601, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=602, scale=0.1)
    shift = hl16_weight_shift(w)
    x16, w16 = to_hl16(x).cuda(), to_hl16(w.double() * 2.0 ** shift).cuda()
    for pool in (False, True):
        Ho = H // 2 if pool else H
        big = torch.full((512 * Ho * Ho, Cout), float('nan')).cuda()
        small = torch.full((2 * Ho * Ho, Cout), float('nan')).cuda()
        hip.conv3x3_hl16_patch(x16, w16, bias.cuda(), big, 512, H, W, Cin, Cout, pool, 2.0 ** -shift)
        hip.conv3x3_hl16_patch(x16[:2 * H * W], w16, bias.cuda(), small, 2, H, W, Cin, Cout, pool, 2.0 ** -shift)
        assert torch.equal(big[:2 * Ho * Ho], small), 'tile width changed the result (pool=%s)' % pool
