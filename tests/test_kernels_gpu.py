"""Per-kernel parity: every C-ABI entry point (called through ctypes on the GPU)
against the executable contract in tests/fake_ops.py (torch fp32/fp64 on CPU).

fp32 MFMA accumulates as an fmaf chain in k order, torch CPU uses a different
summation order, so GEMM-type results agree to ~1e-6 relative; tolerances are
written per test as (rtol on the tensor's max magnitude)."""
import numpy as np
import pytest
import torch

from fake_ops import TorchOps
from mmmot_amd.plan import RowTiles, Segments

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from mmmot_amd.ops import HipOps
    return HipOps()


def G(t):
    return None if t is None else t.cuda()


def close(got, ref, rtol, what):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), '%s: non-finite values' % what
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    if err > rtol * scale:
        idx = np.unravel_index(int((got - ref).abs().argmax()), got.shape)
        nbad = int(((got - ref).abs() > rtol * scale).sum())
        raise AssertionError('%s: max|err| %.3e (scale %.3e, rtol %.1e) at %s got %.6g ref %.6g; %d/%d bad' % (
            what, err, scale, rtol, idx, got[idx].item(), ref[idx].item(), nbad, got.numel()))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


class DevTiles:
    """RowTiles pair: host copy for the emulation, device copy for HIP."""

    def __init__(self, counts):
        self.cpu = RowTiles(counts, 'cpu')
        self.gpu = RowTiles(counts, 'cuda')


@pytest.mark.parametrize('K', [8, 32, 96])
def test_mfma_fragment_layout(hip, K):
    # asymmetric operands: a transposed C write or a swapped A/B mapping cannot pass
    A = (torch.arange(32 * K).float().view(32, K) % 17 - 8) / 8 + rnd(32, K, seed=1) * 0.01
    B = (torch.arange(32 * K).float().view(32, K) % 13 - 3) / 5 + rnd(32, K, seed=2) * 0.01
    C = torch.zeros(32, 32).cuda()
    hip.selftest_mfma(A.cuda(), B.cuda(), C, K)
    close(C, A.double() @ B.double().t(), 2e-6, 'mfma 32x32 K=%d' % K)


CONV_CASES = [
    # first pool L  H   W  Cin Cout
    (1, 0, 3, 8, 8, 3, 64),
    (1, 1, 2, 8, 12, 3, 64),
    (0, 1, 2, 8, 8, 64, 64),
    (0, 0, 2, 8, 12, 64, 128),
    (0, 1, 3, 4, 4, 128, 256),
    (0, 1, 3, 25, 25, 64, 64),   # odd map, floor pooling (nn.MaxPool2d(2, 2)): 25 -> 12
    (0, 0, 2, 5, 7, 32, 64),     # odd, unpooled
    (1, 1, 2, 10, 10, 3, 64),
    (0, 0, 5, 6, 10, 32, 64),
    (0, 1, 7, 2, 2, 256, 512),
    (0, 0, 1, 16, 16, 512, 512),
]


@pytest.mark.parametrize('first,pool,L,H,W,Cin,Cout', CONV_CASES)
def test_conv3x3(hip, first, pool, L, H, W, Cin, Cout):
    emu = TorchOps(torch.float64)
    x = rnd(L, 3, H, W, seed=3) if first else torch.relu(rnd(L, H, W, Cin, seed=3))
    wp = rnd(Cout, 32, seed=4, scale=0.2) if first else rnd(9, Cout, Cin, seed=4, scale=(2.0 / (9 * Cin)) ** 0.5)
    if first:
        wp[:, 27:] = 0
    bias = rnd(Cout, seed=5, scale=0.1)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.zeros(L * Ho * Wo, Cout)
    emu.conv3x3(x, wp, bias, ref, L, H, W, Cin, Cout, first, pool)
    out = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3(x.cuda(), wp.cuda(), bias.cuda(), out, L, H, W, Cin, Cout, first, pool)
    close(out, ref, 3e-6, 'conv3x3')


def _gemm_inputs(counts, N, K, seed):
    R = sum(counts)
    X = rnd(R, K, seed=seed)
    W = rnd(N, K, seed=seed + 1, scale=K ** -0.5)
    bias = rnd(N, seed=seed + 2, scale=0.2)
    return R, X, W, bias


@pytest.mark.parametrize('N,K', [(64, 64), (128, 64), (512, 512), (1024, 128), (256, 512), (64, 1024)])
@pytest.mark.parametrize('counts', [[5, 300, 128], [1], [257]])
def test_gemm_plain_stats(hip, N, K, counts):
    emu = TorchOps(torch.float64)
    tl = DevTiles(counts)
    R, X, W, bias = _gemm_inputs(counts, N, K, 10)
    X = X + 3.0  # non-zero mean: exercises the centred second moment
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    emu.gemm(W, tl.cpu, N, K, X=X, bias=bias, Y=Y, part=part, act=1)
    Yg, pg = torch.full((R, N), float('nan')).cuda(), torch.full((tl.cpu.T, 2, N), float('nan')).cuda()
    hip.gemm(W.cuda(), tl.gpu, N, K, X=X.cuda(), bias=bias.cuda(), Y=Yg, part=pg, act=1)
    close(Yg, Y, 3e-6, 'gemm Y')
    close(pg[:, 0], part[:, 0], 1e-5, 'gemm tile sums')
    close(pg[:, 1], part[:, 1], 1e-4, 'gemm tile M2')


def test_gemm_norm_relu_dbias_strided(hip):
    emu = TorchOps(torch.float64)
    counts = [70, 200]
    tl = DevTiles(counts)
    N, K, R = 128, 64, 270
    buf = rnd(R, 2 * K, seed=20)           # X is a column slice of a wider buffer (ld = 2K)
    X = buf[:, K:]
    W = rnd(N, K, seed=21, scale=K ** -0.5)
    sc, sh = rnd(2, K, seed=22).abs() + 0.5, rnd(2, K, seed=23)
    ndet = 9
    dbias = rnd(ndet, N, seed=24)
    rowidx = (torch.arange(R) * ndet // R).int()
    out = torch.zeros(R, 3 * N)            # Y is a column slice too
    emu.gemm(W, tl.cpu, N, K, X=X, Y=out[:, N:2 * N], sc=sc, sh=sh, amode=1, dbias=dbias, rowidx=rowidx)
    outg = torch.zeros(R, 3 * N).cuda()
    bufg = buf.cuda()
    hip.gemm(W.cuda(), tl.gpu, N, K, X=bufg[:, K:], Y=outg[:, N:2 * N], sc=sc.cuda(), sh=sh.cuda(), amode=1,
             dbias=dbias.cuda(), rowidx=rowidx.cuda())
    close(outg, out, 3e-6, 'gemm norm_relu + dbias (strided views)')


@pytest.mark.parametrize('pairop', [0, 1, 2])
def test_gemm_pair(hip, pairop):
    emu = TorchOps(torch.float64)
    # two groups: 5x7 and 130x3 pairs, features shared in one [rows][512] matrix
    NM = [(5, 7), (130, 3)]
    counts = [n * m for n, m in NM]
    tl = DevTiles(counts)
    K, N = 512, 128
    Fm = rnd(150, K, seed=30)
    W = rnd(N, K, seed=31, scale=K ** -0.5)
    bias = rnd(N, seed=32)
    aoff, boff = [0, 12], [5, 142]
    mk = lambda dev: dict(row0=torch.tensor(tl.cpu.h_g_row0).to(dev), M=torch.tensor([7, 3], dtype=torch.int32).to(dev),
                          aoff=torch.tensor(aoff, dtype=torch.int32).to(dev),
                          boff=torch.tensor(boff, dtype=torch.int32).to(dev))
    R = sum(counts)
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    emu.gemm(W, tl.cpu, N, K, FA=Fm, FB=Fm, pair=mk('cpu'), amode=2, pairop=pairop, bias=bias, Y=Y, part=part)
    Yg, pg = torch.zeros(R, N).cuda(), torch.zeros(tl.cpu.T, 2, N).cuda()
    Fg = Fm.cuda()
    hip.gemm(W.cuda(), tl.gpu, N, K, FA=Fg, FB=Fg, pair=mk('cuda'), amode=2, pairop=pairop, bias=bias.cuda(), Y=Yg,
             part=pg)
    close(Yg, Y, 3e-6, 'gemm pair')
    close(pg[:, 0], part[:, 0], 1e-5, 'pair tile sums')


@pytest.mark.parametrize('C,NG', [(64, 64), (512, 1), (512, 16), (1024, 1024), (128, 128)])
def test_gn_finalize(hip, C, NG):
    emu = TorchOps()
    counts = [300, 5, 128]
    tl = DevTiles(counts)
    g = torch.Generator().manual_seed(40)
    part = torch.zeros(tl.cpu.T, 2, 2 * C)
    n_t = torch.tensor(tl.cpu.h_nrows).float().view(-1, 1)
    part[:, 0] = torch.randn(tl.cpu.T, 2 * C, generator=g) * n_t * 3     # sums with a large mean
    part[:, 1] = torch.rand(tl.cpu.T, 2 * C, generator=g) * n_t * 0.01   # tiny within-tile M2
    gamma, beta = rnd(C, seed=41).abs() + 0.5, rnd(C, seed=42)
    sc, sh = torch.zeros(3, C), torch.zeros(3, C)
    emu.gn_finalize(part[:, :, C:], tl.cpu, C, NG, gamma, beta, 1e-5, sc, sh)
    scg, shg = torch.zeros(3, C).cuda(), torch.zeros(3, C).cuda()
    pg = part.cuda()
    hip.gn_finalize(pg[:, :, C:], tl.gpu, C, NG, gamma.cuda(), beta.cuda(), 1e-5, scg, shg)
    close(scg, sc, 1e-6, 'gn scale')
    close(shg, sh, 1e-6, 'gn shift')


@pytest.mark.parametrize('C', [128, 512, 1024])
def test_segment_mean(hip, C):
    emu = TorchOps()
    R = 400
    X = rnd(R, C + 64, seed=50)[:, 32:32 + C]          # strided view, 16-byte aligned offset
    start = [0, 7, 300, 3, 399]
    count = [7, 293, 100, 50, 1]
    stride = [1, 1, 1, 5, 1]
    group = [0, 1, 0, 1, 1]
    sc, sh = rnd(2, C, seed=51), rnd(2, C, seed=52)
    for norm in (False, True):
        sg_c, sg_g = Segments(start, count, stride, group, 'cpu'), Segments(start, count, stride, group, 'cuda')
        out = torch.zeros(5, C)
        emu.segment_mean(X, C, sg_c, out, sc=sc if norm else None, sh=sh if norm else None, relu=norm)
        Xg = rnd(R, C + 64, seed=50).cuda()[:, 32:32 + C]
        outg = torch.full((5, C), float('nan')).cuda()
        hip.segment_mean(Xg, C, sg_g, outg, sc=G(sc) if norm else None, sh=G(sh) if norm else None, relu=norm)
        close(outg, out, 2e-6, 'segment_mean norm=%s' % norm)


@pytest.mark.parametrize('K', [128, 256, 512])
def test_rowdot(hip, K):
    emu = TorchOps()
    counts = [3, 200, 130]
    tl = DevTiles(counts)
    R = sum(counts)
    X, w = rnd(R, K, seed=60), rnd(K, seed=61, scale=K ** -0.5)
    sc, sh = rnd(3, K, seed=62), rnd(3, K, seed=63)
    omap = torch.randperm(R + 10, generator=torch.Generator().manual_seed(64))[:R].int()
    out = torch.zeros(R + 10)
    emu.rowdot(X, K, w, 0.3, tl.cpu, out, sc=sc, sh=sh, act=2, omap=omap)
    outg = torch.zeros(R + 10).cuda()
    hip.rowdot(X.cuda(), K, w.cuda(), 0.3, tl.gpu, outg, sc=sc.cuda(), sh=sh.cuda(), act=2, omap=omap.cuda())
    close(outg, out, 2e-6, 'rowdot norm+sigmoid+omap')
    out2 = torch.zeros(R)
    emu.rowdot(X, K, w, -0.1, tl.cpu, out2, act=2, use_thr=True, thr=0.45)
    outg2 = torch.zeros(R).cuda()
    hip.rowdot(X.cuda(), K, w.cuda(), -0.1, tl.gpu, outg2, act=2, use_thr=True, thr=0.45)
    # values within 1e-6 of the threshold may legitimately flip
    safe = (torch.sigmoid(X @ w - 0.1) - 0.45).abs() > 1e-5
    close(outg2.cpu()[safe], out2[safe], 2e-6, 'rowdot threshold')


@pytest.mark.parametrize('C', [64, 128, 512])
def test_row_layernorm(hip, C):
    emu = TorchOps()
    R = 37
    X = rnd(R, C, seed=70) * 3 + 5
    gamma, beta = rnd(C, seed=71), rnd(C, seed=72)
    Y = torch.zeros(R, 2 * C)
    emu.row_layernorm(X, C, gamma, beta, 1e-5, True, Y[:, C:], R)
    Yg = torch.zeros(R, 2 * C).cuda()
    hip.row_layernorm(X.cuda(), C, gamma.cuda(), beta.cuda(), 1e-5, True, Yg[:, C:], R)
    close(Yg, Y, 3e-6, 'row_layernorm')


@pytest.mark.parametrize('K', [3, 4])
def test_pointnet_layer1(hip, K):
    emu = TorchOps()
    counts = [1000, 3, 129]
    tl = DevTiles(counts)
    R = sum(counts)
    X = rnd(R, K, seed=80) * 10 + torch.tensor([30.0, -5.0, -1.0, 0.5][:K])
    W, b = rnd(64, K, seed=81), rnd(64, seed=82)
    Y, part = torch.zeros(R, 64), torch.zeros(tl.cpu.T, 2, 64)
    emu.pointnet_layer1(X, W, b, Y, part, tl.cpu)
    Yg, pg = torch.zeros(R, 64).cuda(), torch.zeros(tl.cpu.T, 2, 64).cuda()
    hip.pointnet_layer1(X.cuda(), W.cuda(), b.cuda(), Yg, pg, tl.gpu)
    close(Yg, Y, 2e-6, 'pointnet layer1')
    close(pg[:, 0], part[:, 0], 1e-5, 'layer1 sums')
    close(pg[:, 1], part[:, 1], 1e-4, 'layer1 M2')


def test_affine_act(hip):
    emu = TorchOps()
    counts = [40, 130]
    tl = DevTiles(counts)
    C, R = 512, 170
    X, sc, sh = rnd(R, C, seed=90), rnd(2, C, seed=91), rnd(2, C, seed=92)
    Y = torch.zeros(R, 2 * C)
    emu.affine_act(X, C, sc, sh, tl.cpu, 1, Y[:, C:])
    Yg = torch.zeros(R, 2 * C).cuda()
    hip.affine_act(X.cuda(), C, sc.cuda(), sh.cuda(), tl.gpu, 1, Yg[:, C:])
    close(Yg, Y, 1e-6, 'affine_act')


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_fusion_combine(hip, mode):
    emu = TorchOps()
    counts = [12, 140]
    tl = DevTiles(counts)
    C, Lt = 512, 152
    N = 2 * C if mode == 2 else C
    cat = rnd(Lt, 2 * C, seed=100)
    Y0, Y1 = rnd(Lt, N, seed=101), rnd(Lt, N, seed=102)
    s0, h0, s1, h1 = (rnd(2, N, seed=103 + i) for i in range(4))
    v = lambda t: t[:, N - C:]
    F = torch.zeros(3, Lt, C)
    emu.fusion_combine(mode, cat, Y0, Y1 if mode else None, v(s0), v(h0), v(s1) if mode else None,
                       v(h1) if mode else None, tl.cpu, F, Lt, C)
    Fg = torch.zeros(3, Lt, C).cuda()
    d = lambda t: t.cuda()
    s0g, h0g, s1g, h1g = d(s0), d(h0), d(s1), d(h1)
    hip.fusion_combine(mode, d(cat), d(Y0), d(Y1) if mode else None, v(s0g), v(h0g), v(s1g) if mode else None,
                       v(h1g) if mode else None, tl.gpu, Fg, Lt, C)
    close(Fg, F, 3e-6, 'fusion_combine mode %d' % mode)


@pytest.mark.parametrize('mode', [1, 2, 3, 4])
def test_softmax_pairs(hip, mode):
    emu = TorchOps()
    NM = [(5, 7), (1, 1), (1, 9), (130, 70), (64, 1)]
    row0 = np.concatenate([[0], np.cumsum([n * m for n, m in NM])]).astype(np.int32)
    R = int(row0[-1])
    x = rnd(R, seed=110) * 4
    r0 = torch.tensor(row0[:-1])
    gN, gM = torch.tensor([n for n, _ in NM], dtype=torch.int32), torch.tensor([m for _, m in NM], dtype=torch.int32)
    out = torch.zeros(R)
    emu.softmax_pairs(x, out, r0, gN, gM, len(NM), 200, mode)
    outg = torch.zeros(R).cuda()
    hip.softmax_pairs(x.cuda(), outg, r0.cuda(), gN.cuda(), gM.cuda(), len(NM), 200, mode)
    close(outg, out, 3e-6, 'softmax mode %d' % mode)


def test_cpu_tensors_rejected(hip):
    with pytest.raises(RuntimeError):
        hip.selftest_mfma(torch.zeros(32, 8), torch.zeros(32, 8), torch.zeros(32, 32), 8)


# ---- fp16-split (hl16) trunk ---------------------------------------------------------------
def test_hl16_pack_unpack_matches_host_format(hip):
    from mmmot_amd.pack import from_hl16, to_hl16
    x = torch.cat([rnd(4096, seed=120) * 30, rnd(4096, seed=121) * 1e-3, rnd(64, seed=122) * 1e-6])
    y = torch.zeros_like(x).cuda()
    hip.hl16_pack(x.cuda(), y)
    assert torch.equal(y.cpu().view(torch.int32), to_hl16(x.view(-1, 8)).reshape(-1).view(torch.int32))
    z = torch.zeros_like(x).cuda()
    hip.hl16_unpack(y, z)
    close(z, from_hl16(to_hl16(x.view(-1, 8))).reshape(-1), 1e-7, 'hl16 unpack')
    # 22 significand bits: relative error <= 2^-21 for normal-range values
    big = x[:4096].double()
    assert ((z.cpu()[:4096].double() - big).abs() <= big.abs() * 2.0 ** -21 + 1e-7).all()


HL_CASES = [
    # pool L  H   W  Cin Cout
    (1, 2, 8, 8, 64, 64),
    (0, 2, 8, 12, 64, 128),
    (1, 3, 4, 4, 128, 256),
    (0, 5, 6, 10, 64, 64),
    (1, 7, 2, 2, 256, 512),
    (0, 1, 16, 16, 512, 512),
    (1, 9, 4, 4, 512, 512),     # 36 pooled pixels: partial tile + channel-tile-major order
]


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', HL_CASES)
def test_conv3x3_hl16(hip, pool, L, H, W, Cin, Cout):
    from mmmot_amd.pack import from_hl16, hl16_weight_shift, to_hl16
    x = torch.relu(rnd(L * H * W, Cin, seed=130)) * 3.0
    w = rnd(9, Cout, Cin, seed=131, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=132, scale=0.1)
    shift = hl16_weight_shift(w)
    x16, w16 = to_hl16(x), to_hl16(w.double() * 2.0 ** shift)
    # reference: fp64 conv on the values the kernel actually sees (hi + lo), then the output split rounding
    emu = TorchOps(torch.float64)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.zeros(L * Ho * Wo, Cout)
    emu.conv3x3(from_hl16(x16).view(L, H, W, Cin), (from_hl16(w16) * 2.0 ** -shift), bias, ref, L, H, W, Cin, Cout,
                False, bool(pool))
    out16 = torch.zeros(L * Ho * Wo, Cout).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), out16, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    close(out, ref, 2e-6, 'conv3x3 hl16 (3-term fp16 split) vs fp64')


def test_conv3x3_first_hl16(hip):
    emu = TorchOps(torch.float64)
    L, H, W, Cout = 3, 8, 12, 64
    x = rnd(L, 3, H, W, seed=140)
    wp = rnd(Cout, 32, seed=141, scale=0.2)
    wp[:, 27:] = 0
    bias = rnd(Cout, seed=142, scale=0.1)
    ref = torch.zeros(L * H * W, Cout)
    emu.conv3x3(x, wp, bias, ref, L, H, W, 3, Cout, True, False)
    out16 = torch.zeros(L * H * W, Cout).cuda()
    hip.conv3x3_first_hl16(x.cuda(), wp.cuda(), bias.cuda(), out16, L, H, W, Cout)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    close(out, ref, 3e-6, 'first conv -> hl16')


def test_hl16_small_magnitudes_keep_absolute_accuracy(hip):
    """Small activations put their lo halves into fp16's subnormal range; the result must stay
    accurate in ABSOLUTE terms (the MFMA must not flush fp16 subnormal inputs to zero)."""
    from mmmot_amd.pack import from_hl16, hl16_weight_shift, to_hl16
    L, H, W, Cin, Cout = 1, 8, 8, 64, 64
    x = torch.relu(rnd(L * H * W, Cin, seed=150)) * 0.02          # lo halves ~1e-6: fp16 subnormals
    w = rnd(9, Cout, Cin, seed=151, scale=0.05)
    bias = torch.zeros(Cout)
    shift = hl16_weight_shift(w)
    x16, w16 = to_hl16(x), to_hl16(w.double() * 2.0 ** shift)
    emu = TorchOps(torch.float64)
    ref = torch.zeros(L * H * W, Cout)
    emu.conv3x3(x.view(L, H, W, Cin), w, bias, ref, L, H, W, Cin, Cout, False, False)   # exact fp32 inputs
    out16 = torch.zeros(L * H * W, Cout).cuda()
    hip.conv3x3_hl16_patch(x16.cuda(), w16.cuda(), bias.cuda(), out16, L, H, W, Cin, Cout, False, 2.0 ** -shift)
    out = torch.zeros_like(out16)
    hip.hl16_unpack(out16, out)
    err = (out.cpu().double() - ref.double()).abs().max().item()
    print('hl16 small-magnitude abs err %.3e (ref max %.3e)' % (err, ref.abs().max().item()))
    assert err < 2e-6, err   # a flushed lo half would cost ~2^-11 relative = 1e-4 here


@pytest.mark.parametrize('C', [64, 128, 256, 512])
def test_segment_mean_hl16_input(hip, C):
    """hl16 rows: C = 128 / 256 / 512 without a prologue take the unit-per-lane kernel (16 / 32 / 64 lanes per row), other
    shapes the general one; ragged counts around the kernel's pass sizes (4 x 64 / UPR rows per workgroup pass, 4 passes
    unrolled), a strided segment and the explicit divisor of the two-level crop pool."""
    from mmmot_amd.pack import from_hl16, to_hl16
    emu = TorchOps()
    counts = [1, 3, 16, 17, 63, 64, 65, 255, 256, 300, 2]
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    R = int(sum(counts)) + 600
    x = torch.relu(rnd(R, C, seed=160)) * 4
    x16 = to_hl16(x)
    # one strided segment on the spare rows: every second row
    starts = np.concatenate([starts, [sum(counts)]])
    cnts = np.array(counts + [250])
    strides = np.array([1] * len(counts) + [2])
    n = len(cnts)
    sg_c = Segments(starts, cnts, strides, np.zeros(n), 'cpu')
    sg_g = Segments(starts, cnts, strides, np.zeros(n), 'cuda')
    out = torch.zeros(n, C)
    emu.segment_mean(from_hl16(x16), C, sg_c, out, use_group=False)
    outg = torch.full((n, C), float('nan')).cuda()
    hip.segment_mean(x16.cuda(), C, sg_g, outg, use_group=False, hl16=True)
    close(outg, out, 2e-6, 'segment_mean on hl16 rows')
    # explicit divisor (first level of the two-level pool: partial sums = mean * count / div)
    sg_c = Segments(starts, cnts, strides, np.zeros(n), 'cpu', div=np.full(n, 7))
    sg_g = Segments(starts, cnts, strides, np.zeros(n), 'cuda', div=np.full(n, 7))
    emu.segment_mean(from_hl16(x16), C, sg_c, out, use_group=False)
    outg.fill_(float('nan'))
    hip.segment_mean(x16.cuda(), C, sg_g, outg, use_group=False, hl16=True)
    close(outg, out, 2e-6, 'segment_mean on hl16 rows with a divisor')


# ---- fused "next GroupNorm + ReLU + per-detection mean" path (C-ABI v2) -----------------------------------
@pytest.mark.parametrize('w_hl16', [False, True])
def test_gemm_colsum_on_ragged_tiles_end_to_end(hip, w_hl16):
    """Two GEMM passes (statistics only, then normalise+ReLU+column sums) on detection-aligned tiles followed
    by the tile-segment mean must equal: materialise Y, GroupNorm(C,C), ReLU, per-detection mean."""
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    emu = TorchOps(torch.float64)
    det_counts = [[130, 1, 40, 257], [5, 128, 300]]          # two samples, ragged detections
    counts = [sum(d) for d in det_counts]
    cpu = RowTiles(counts, 'cpu', sub_counts=det_counts)
    gpu = RowTiles(counts, 'cuda', sub_counts=det_counts)
    assert cpu.ragged and (cpu.h_nrows < 128).sum() > 2
    ndet = sum(len(d) for d in det_counts)
    flat = [c for d in det_counts for c in d]
    N, K, R = 256, 128, sum(counts)
    X = rnd(R, K, seed=70)
    W = rnd(N, K, seed=71, scale=K ** -0.5)
    bias = rnd(N, seed=72)
    sc, sh = rnd(2, K, seed=73).abs() + 0.5, rnd(2, K, seed=74)
    gamma, beta = rnd(N, seed=75).abs() + 0.5, rnd(N, seed=76)
    dbias = rnd(ndet, N, seed=77)
    rowidx = torch.repeat_interleave(torch.arange(ndet), torch.tensor(flat)).int()
    # reference: materialised path in fp64
    Y = torch.zeros(R, N, dtype=torch.float64)
    plain = RowTiles(counts, 'cpu')
    part = torch.zeros(plain.T, 2, N, dtype=torch.float64)
    emu.gemm(W, plain, N, K, X=X, bias=bias, Y=Y, part=part, sc=sc, sh=sh, amode=1, dbias=dbias, rowidx=rowidx)
    osc_r, osh_r = torch.zeros(2, N), torch.zeros(2, N)
    emu.gn_finalize(part, plain, N, N, gamma, beta, 1e-5, osc_r, osh_r)
    starts = torch.tensor([0] + flat).cumsum(0)[:-1].tolist()
    grp = [0] * len(det_counts[0]) + [1] * len(det_counts[1])
    ref = torch.zeros(ndet, N, dtype=torch.float64)
    emu.segment_mean(Y, N, Segments(starts, flat, [1] * ndet, grp, 'cpu'), ref, sc=osc_r.double(), sh=osh_r.double(),
                     relu=True)
    # HIP: fused path
    if w_hl16:
        shift = hl16_weight_shift(W)
        Wg, kw = to_hl16(W.double() * 2.0 ** shift).cuda(), dict(w_hl16=True, oscale=2.0 ** -shift)
    else:
        Wg, kw = W.cuda(), {}
    args = dict(X=X.cuda(), bias=bias.cuda(), sc=sc.cuda(), sh=sh.cuda(), amode=1, dbias=dbias.cuda(),
                rowidx=rowidx.cuda(), **kw)
    pg = torch.full((gpu.T, 2, N), float('nan')).cuda()
    hip.gemm(Wg, gpu, N, K, part=pg, **args)
    oscg, oshg = torch.zeros(2, N).cuda(), torch.zeros(2, N).cuda()
    hip.gn_finalize(pg, gpu, N, N, gamma.cuda(), beta.cuda(), 1e-5, oscg, oshg)
    close(oscg, osc_r, 2e-5, 'scale from ragged-tile statistics')
    close(oshg, osh_r, 2e-5, 'shift from ragged-tile statistics')
    cs = torch.full((gpu.T, N), float('nan')).cuda()
    hip.gemm(Wg, gpu, N, K, osc=oscg, osh=oshg, colsum=cs, **args)
    segs = Segments(cpu.h_sub_tile0, cpu.h_sub_ntiles, [1] * ndet, grp, 'cuda', div=flat)
    out = torch.full((ndet, N), float('nan')).cuda()
    hip.segment_mean(cs, N, segs, out)
    close(out, ref.float(), 2e-5, 'fused norm+relu+segment mean')
    # and the emulation of the same calls (documents the contract of colsum / seg_div)
    pe, ce, oe = torch.zeros(cpu.T, 2, N), torch.zeros(cpu.T, N), torch.zeros(ndet, N)
    e32 = TorchOps(torch.float64)
    e32.gemm(W, cpu, N, K, X=X, bias=bias, part=pe, sc=sc, sh=sh, amode=1, dbias=dbias, rowidx=rowidx)
    se, he = torch.zeros(2, N), torch.zeros(2, N)
    e32.gn_finalize(pe, cpu, N, N, gamma, beta, 1e-5, se, he)
    e32.gemm(W, cpu, N, K, X=X, bias=bias, sc=sc, sh=sh, amode=1, dbias=dbias, rowidx=rowidx, osc=se, osh=he, colsum=ce)
    e32.segment_mean(ce, N, Segments(cpu.h_sub_tile0, cpu.h_sub_ntiles, [1] * ndet, grp, 'cpu', div=flat), oe)
    close(oe, ref.float(), 2e-5, 'emulated fused path')


@pytest.mark.parametrize('K,N', [(128, 1024), (64, 512), (64, 128)])
def test_gemm_ares_statistics_and_colsum(hip, K, N):
    """A-resident GEMM: per-half-tile statistics and normalise+ReLU column sums vs the fp64 emulation, on
    detection-aligned tiles with empty second halves (<= 64 rows), partial halves and a per-tile bias row."""
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    from mmmot_amd.plan import HalfTiles
    emu = TorchOps(torch.float64)
    det_counts = [[130, 1, 40, 257, 64], [5, 128, 300, 65]]
    counts = [sum(d) for d in det_counts]
    cpu = RowTiles(counts, 'cpu', sub_counts=det_counts)
    gpu = RowTiles(counts, 'cuda', sub_counts=det_counts)
    hc, hg = HalfTiles(cpu, 'cpu'), HalfTiles(gpu, 'cuda')
    flat = [c for d in det_counts for c in d]
    ndet, R = len(flat), sum(counts)
    X = rnd(R, K, seed=80) + 0.5
    W = rnd(N, K, seed=81, scale=K ** -0.5)
    bias = rnd(N, seed=82)
    sc, sh = rnd(2, K, seed=83).abs() + 0.5, rnd(2, K, seed=84)
    dbias = rnd(ndet, N, seed=85)
    tile_det = torch.repeat_interleave(torch.arange(ndet), torch.tensor(cpu.h_sub_ntiles).long()).int()
    osc, osh = rnd(2, N, seed=86).abs() + 0.5, rnd(2, N, seed=87)
    shift = hl16_weight_shift(W)
    W16 = to_hl16(W.double() * 2.0 ** shift)
    osv = 2.0 ** -shift
    part, cs = torch.zeros(hc.T, 2, N, dtype=torch.float64), torch.zeros(hc.T, N, dtype=torch.float64)
    emu.gemm_ares(W16, osv, cpu, N, K, X, sc, sh, bias=bias, dbias=dbias, tile_dbrow=tile_det, part=part,
                  osc=osc, osh=osh, colsum=cs)
    pg, cg = torch.full((hg.T, 2, N), float('nan')).cuda(), torch.full((hg.T, N), float('nan')).cuda()
    hip.gemm_ares(W16.cuda(), osv, gpu, N, K, X.cuda(), sc.cuda(), sh.cuda(), bias=bias.cuda(), dbias=dbias.cuda(),
                  tile_dbrow=tile_det.cuda(), part=pg, osc=osc.cuda(), osh=osh.cuda(), colsum=cg)
    close(pg[:, 0], part[:, 0].float(), 1e-5, 'ares half-tile sums')
    close(pg[:, 1], part[:, 1].float(), 1e-4, 'ares half-tile M2')
    close(cg, cs.float(), 1e-5, 'ares column sums')
    # the statistics feed gn_finalize through the half-tile tables
    gamma, beta = rnd(N, seed=88).abs() + 0.5, rnd(N, seed=89)
    s_r, h_r = torch.zeros(2, N), torch.zeros(2, N)
    emu.gn_finalize(part, hc, N, N, gamma, beta, 1e-5, s_r, h_r)
    s_g, h_g = torch.zeros(2, N).cuda(), torch.zeros(2, N).cuda()
    hip.gn_finalize(pg, hg, N, N, gamma.cuda(), beta.cuda(), 1e-5, s_g, h_g)
    close(s_g, s_r, 2e-5, 'scale from half-tile statistics')
    close(h_g, h_r, 2e-5, 'shift from half-tile statistics')


@pytest.mark.parametrize('N,dets,with_dbias', [
    (1024, [[130, 1, 40, 257, 64], [5, 128, 300, 65]], True),           # ragged detection-aligned tiles, per-tile bias row
    (1024, [[2048] * 9, [2048] * 7, [100]], False),                      # conv5's shape: full tiles, several per sequence
    (512, [[64] * 40], False),                                           # one channel half; every second half tile empty
    (1536, [[700, 33], [128]], True),                                    # three channel halves
    (1024, [[2048] * 40, [300] * 30, [31, 95, 129]], True),              # five and more tiles per workgroup, ragged tails mid-sequence
])
def test_gemm_ares_weights_in_registers_kernel(hip, N, dets, with_dbias):
    """the K = 128 consumer pass on the weights-in-registers kernel (csrc/gemm_wreg.hip): fp64 emulation, and the streaming
    kernel's column sums bit for bit (same operand values, accumulation order and summation order)"""
    from mmmot_amd import _lib
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    from mmmot_amd.plan import HalfTiles
    lib = _lib.load()
    emu = TorchOps(torch.float64)
    K = 128
    counts = [sum(d) for d in dets]
    G = len(dets)
    cpu = RowTiles(counts, 'cpu', sub_counts=dets)
    gpu = RowTiles(counts, 'cuda', sub_counts=dets)
    hc = HalfTiles(cpu, 'cpu')
    flat = [c for d in dets for c in d]
    ndet, R = len(flat), sum(counts)
    X = rnd(R, K, seed=90) + 0.5
    W = rnd(N, K, seed=91, scale=K ** -0.5)
    bias = rnd(N, seed=92)
    sc, sh = rnd(G, K, seed=93).abs() + 0.5, rnd(G, K, seed=94)
    dbias = rnd(ndet, N, seed=95) if with_dbias else None
    tile_det = torch.repeat_interleave(torch.arange(ndet), torch.tensor(cpu.h_sub_ntiles).long()).int()
    osc, osh = rnd(G, N, seed=96).abs() + 0.5, rnd(G, N, seed=97)
    shift = hl16_weight_shift(W)
    W16 = to_hl16(W.double() * 2.0 ** shift)
    osv = 2.0 ** -shift
    cs = torch.zeros(hc.T, N, dtype=torch.float64)
    emu.gemm_ares(W16, osv, cpu, N, K, X, sc, sh, bias=bias, dbias=dbias, tile_dbrow=tile_det if with_dbias else None,
                  osc=osc, osh=osh, colsum=cs)
    outs = {}
    try:
        for v in (2, 1):
            assert lib.mmmot_set_gemm_ares_variant(v) == 0
            cg = torch.full((hc.T, N), float('nan')).cuda()
            hip.gemm_ares(W16.cuda(), osv, gpu, N, K, X.cuda(), sc.cuda(), sh.cuda(), bias=bias.cuda(),
                          dbias=dbias.cuda() if with_dbias else None, tile_dbrow=tile_det.cuda() if with_dbias else None,
                          osc=osc.cuda(), osh=osh.cuda(), colsum=cg)
            outs[v] = cg.cpu()
    finally:
        assert lib.mmmot_set_gemm_ares_variant(0) == 0
    close(outs[2], cs.float(), 1e-5, 'weights-in-registers kernel: column sums')
    if N % 256 == 0:  # the streaming kernel walks 256-channel tiles
        assert torch.equal(outs[2], outs[1]), 'register-resident and streaming kernels differ'


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5, 6, 7])
def test_gemm_ares_k64_independent_wave_kernel_bitwise(hip, seed):
    """the K = 64 column-sum pass on the independent-wave kernel (csrc/gemm_wres.hip, gemm_wres64i_kernel: a wave owns a
    64-row half tile for every channel block, A fragments straight from global loads) against the two-barrier
    weight-resident kernel and the streaming kernel: random detection sizes (ragged tiles, empty second halves, one-row
    tiles; seed 7: more waves than half tiles), with and without per-detection bias rows, N = 128 .. 512 - bitwise, twice;
    and against the float64 emulation"""
    from mmmot_amd import _lib
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    from mmmot_amd.plan import HalfTiles
    lib = _lib.load()
    rng = np.random.default_rng(2000 + seed)
    K = 64
    N = int(rng.choice([128, 256, 384, 512]))
    G = int(rng.integers(1, 4))
    sizes = [1, 31, 32, 33, 64, 65, 127, 128, 129, 300, 1000, 2048, 3000]
    ndets = int(rng.integers(3, 40)) if seed != 7 else 2
    dets = [[int(v) for v in rng.choice(sizes, size=ndets)] for _ in range(G)]
    if seed in (2, 3):  # long runs of full tiles: every wave walks several items, the table changes mid-run
        dets = [d + [2048] * 150 for d in dets]
    with_dbias = bool(seed & 1)
    counts = [sum(d) for d in dets]
    gpu = RowTiles(counts, 'cuda', sub_counts=dets)
    cpu = RowTiles(counts, 'cpu', sub_counts=dets)
    hc = HalfTiles(cpu, 'cpu')
    flat = [c for d in dets for c in d]
    ndet, R = len(flat), sum(counts)
    X = rnd(R, K, seed=400 + seed) + 0.5
    W = rnd(N, K, seed=401 + seed, scale=K ** -0.5)
    bias = rnd(N, seed=402 + seed)
    sc, sh = rnd(G, K, seed=403 + seed).abs() + 0.5, rnd(G, K, seed=404 + seed)
    dbias = rnd(ndet, N, seed=405 + seed) if with_dbias else None
    tile_det = torch.repeat_interleave(torch.arange(ndet), torch.tensor(cpu.h_sub_ntiles).long()).int()
    osc, osh = rnd(G, N, seed=406 + seed).abs() + 0.5, rnd(G, N, seed=407 + seed)
    shift = hl16_weight_shift(W)
    W16 = to_hl16(W.double() * 2.0 ** shift)
    outs = []
    try:
        for v in (3, 2, 2, 1, 0):
            assert lib.mmmot_set_gemm_ares_variant(v) == 0
            cg = torch.full((hc.T, N), float('nan')).cuda()
            hip.gemm_ares(W16.cuda(), 2.0 ** -shift, gpu, N, K, X.cuda(), sc.cuda(), sh.cuda(), bias=bias.cuda(),
                          dbias=dbias.cuda() if with_dbias else None, tile_dbrow=tile_det.cuda() if with_dbias else None,
                          osc=osc.cuda(), osh=osh.cuda(), colsum=cg)
            outs.append(cg.cpu())
    finally:
        assert lib.mmmot_set_gemm_ares_variant(0) == 0
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[1], outs[2]), 'independent-wave kernel: two launches differ'
    assert torch.equal(outs[1], outs[0]), 'independent-wave and two-barrier kernels differ'
    assert torch.equal(outs[1], outs[4]), 'automatic choice differs'
    if N % 256 == 0:
        assert torch.equal(outs[1], outs[3]), 'independent-wave and streaming kernels differ'
    if seed < 2:
        emu = TorchOps(torch.float64)
        cs = torch.zeros(hc.T, N, dtype=torch.float64)
        emu.gemm_ares(W16, 2.0 ** -shift, cpu, N, K, X, sc, sh, bias=bias, dbias=dbias,
                      tile_dbrow=tile_det if with_dbias else None, osc=osc, osh=osh, colsum=cs)
        close(outs[1], cs.float(), 1e-5, 'independent-wave kernel: column sums')


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_gemm_ares_register_kernel_random_tilings_bitwise(hip, seed):
    """the pipelined register-weights kernel against the streaming kernel on random detection sizes (1 .. 3000 points:
    ragged tiles at every position of a workgroup's tile sequence, empty second halves, one-row tiles), random sample
    sizes, with and without per-detection bias rows, N = 512 / 1024 / 2048 - bitwise, twice (the kernel's LDS-DMA
    pipeline has no data-dependent timing in its results)"""
    from mmmot_amd import _lib
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    from mmmot_amd.plan import HalfTiles
    lib = _lib.load()
    rng = np.random.default_rng(1000 + seed)
    K = 128
    N = int(rng.choice([512, 1024, 2048]))
    G = int(rng.integers(1, 4))
    dets = [[int(v) for v in rng.choice([1, 31, 32, 33, 64, 65, 127, 128, 129, 300, 1000, 2048, 3000],
                                        size=int(rng.integers(3, 40)))] for _ in range(G)]
    with_dbias = bool(seed & 1)
    counts = [sum(d) for d in dets]
    gpu = RowTiles(counts, 'cuda', sub_counts=dets)
    hc = HalfTiles(RowTiles(counts, 'cpu', sub_counts=dets), 'cpu')
    flat = [c for d in dets for c in d]
    ndet, R = len(flat), sum(counts)
    X = (rnd(R, K, seed=300 + seed) + 0.5).cuda()
    W = rnd(N, K, seed=301 + seed, scale=K ** -0.5)
    bias = rnd(N, seed=302 + seed).cuda()
    sc, sh = (rnd(G, K, seed=303 + seed).abs() + 0.5).cuda(), rnd(G, K, seed=304 + seed).cuda()
    dbias = rnd(ndet, N, seed=305 + seed).cuda() if with_dbias else None
    cpu_tiles = RowTiles(counts, 'cpu', sub_counts=dets)
    tile_det = torch.repeat_interleave(torch.arange(ndet), torch.tensor(cpu_tiles.h_sub_ntiles).long()).int().cuda()
    osc, osh = (rnd(G, N, seed=306 + seed).abs() + 0.5).cuda(), rnd(G, N, seed=307 + seed).cuda()
    shift = hl16_weight_shift(W)
    W16 = to_hl16(W.double() * 2.0 ** shift).cuda()
    outs = []
    try:
        for v in (1, 2, 2):
            assert lib.mmmot_set_gemm_ares_variant(v) == 0
            cg = torch.full((hc.T, N), float('nan')).cuda()
            hip.gemm_ares(W16, 2.0 ** -shift, gpu, N, K, X, sc, sh, bias=bias, dbias=dbias,
                          tile_dbrow=tile_det if with_dbias else None, osc=osc, osh=osh, colsum=cg)
            outs.append(cg.cpu())
    finally:
        assert lib.mmmot_set_gemm_ares_variant(0) == 0
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[1], outs[2]), 'register kernel: two launches differ'
    if N % 256 == 0:
        assert torch.equal(outs[1], outs[0]), 'register-resident and streaming kernels differ'


@pytest.mark.parametrize('K,N', [(128, 1024), (64, 192)])
def test_gram_statistics_match_direct_statistics(hip, K, N):
    """GroupNorm(N, N) scale/shift of v = W relu(X*sc+sh) + b from the Gram matrix of the input (gram_rows +
    gn_finalize_gram) vs float64 statistics of v itself; ragged super-tiles, two groups, |mean| >> std inputs."""
    emu = TorchOps(torch.float64)
    counts = [3000, 1025]
    cpu, gpu = RowTiles(counts, 'cpu', tile=1024), RowTiles(counts, 'cuda', tile=1024)
    assert cpu.T == 5 and int(cpu.h_nrows.min()) == 1
    R = sum(counts)
    X = rnd(R, K, seed=90) * 2.0 + 0.7
    sc, sh = rnd(2, K, seed=91).abs() + 0.5, rnd(2, K, seed=92) * 0.5
    W = rnd(N, K, seed=93, scale=K ** -0.5)
    bias, gamma, beta = rnd(N, seed=94), rnd(N, seed=95).abs() + 0.5, rnd(N, seed=96)
    grp = torch.repeat_interleave(torch.arange(2), torch.tensor(counts))
    A = torch.relu(X.double() * sc.double()[grp] + sh.double()[grp])
    v = A @ W.double().t() + bias.double()
    sc_ref, sh_ref = torch.zeros(2, N, dtype=torch.float64), torch.zeros(2, N, dtype=torch.float64)
    for g in range(2):
        vg = v[grp == g]
        s = gamma.double() / torch.sqrt(vg.var(0, unbiased=False) + 1e-5)
        sc_ref[g], sh_ref[g] = s, beta.double() - vg.mean(0) * s
    Gp = torch.zeros(gpu.T, K * K, dtype=torch.float64).cuda()
    Sp = torch.zeros(gpu.T, K, dtype=torch.float64).cuda()
    hip.gram_rows(X.cuda(), K, sc.cuda(), sh.cuda(), gpu, Gp, Sp)
    Ge, Se = torch.zeros(cpu.T, K * K, dtype=torch.float64), torch.zeros(cpu.T, K, dtype=torch.float64)
    emu.gram_rows(X, K, sc, sh, cpu, Ge, Se)
    close(Sp.float(), Se.float(), 2e-6, 'column sums')
    if K == 128:  # the K = 128 kernel writes the blocks on and above the diagonal of the 4 x 4 grid of 32 x 32 blocks
        blk = torch.arange(K) // 32
        upper = (blk[:, None] <= blk[None, :]).reshape(1, K * K).cuda()
        Gm = Gp.view(-1, K, K)
        Gp_full = torch.where(upper.view(1, K, K), Gm, Gm.transpose(1, 2)).reshape(-1, K * K)
        assert torch.equal(Gp[:, upper[0]], Gp_full[:, upper[0]]) and bool((Gp[:, ~upper[0]] == 0).all()), 'lower blocks are not written'
    else:
        Gp_full = Gp
    close(Gp_full.float(), Ge.float(), 2e-6, 'Gram partials')
    if K == 128:  # the two forms of the K = 128 kernel (four waves x two workgroups per CU / eight waves, pipelined): same bits
        from mmmot_amd import _lib
        lib = _lib.load()
        outs = []
        for v in (1, 2):
            assert lib.mmmot_set_gram128_variant(v) == 0
            G2, S2 = torch.zeros_like(Gp), torch.zeros_like(Sp)
            hip.gram_rows(X.cuda(), K, sc.cuda(), sh.cuda(), gpu, G2, S2)
            outs.append((G2, S2))
        assert lib.mmmot_set_gram128_variant(0) == 0
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    scg, shg = torch.zeros(2, N).cuda(), torch.zeros(2, N).cuda()
    work = torch.zeros(2, K * K + K, dtype=torch.float64).cuda()
    hip.gn_finalize_gram(Gp, Sp, gpu, K, W.cuda(), bias.cuda(), N, gamma.cuda(), beta.cuda(), 1e-5, work, scg, shg)
    close(scg, sc_ref.float(), 5e-6, 'scale from the Gram route')
    close(shg, sh_ref.float(), 5e-6, 'shift from the Gram route')


@pytest.mark.parametrize('dets,tile', [([[700, 1, 130, 2048], [90, 300]], 512), ([[3, 5, 1]], 128), ([[2048] * 6, [1024] * 5], 2048)])
def test_gram_statistics_with_a_gathered_per_detection_bias(hip, dets, tile):
    """GroupNorm(N, N) scale/shift of v = W relu(X*sc+sh) + dbias[det] (PointNet_v1.conv1 after the 1088 -> 512 split) from
    the Gram partials of the 64-channel input over detection-aligned super-tiles (gram_rows K = 64 +
    gn_finalize_gram_dbias) against float64 statistics of v itself; ragged detections incl. one-point ones, several
    super-tiles per detection, two groups, a bias whose spread across detections exceeds the spread of W a."""
    emu = TorchOps(torch.float64)
    K, N = 64, 512
    counts = [sum(d) for d in dets]
    G = len(dets)
    cpu = RowTiles(counts, 'cpu', sub_counts=dets, tile=tile)
    gpu = RowTiles(counts, 'cuda', sub_counts=dets, tile=tile)
    Lt = sum(len(d) for d in dets)
    tile_det = torch.from_numpy(np.repeat(np.arange(Lt), cpu.h_sub_ntiles)).int()
    R = sum(counts)
    X = rnd(R, K, seed=190) * 2.0 + 0.7
    sc, sh = rnd(G, K, seed=191).abs() + 0.5, rnd(G, K, seed=192) * 0.5
    W = rnd(N, K, seed=193, scale=K ** -0.5)
    dbias = rnd(Lt, N, seed=194) * 3.0 + 1.5
    gamma, beta = rnd(N, seed=195).abs() + 0.5, rnd(N, seed=196)
    grp = torch.repeat_interleave(torch.arange(G), torch.tensor(counts))
    row_det = torch.repeat_interleave(torch.arange(Lt), torch.tensor([c for d in dets for c in d]))
    A = torch.relu(X.double() * sc.double()[grp] + sh.double()[grp])
    v = A @ W.double().t() + dbias.double()[row_det]
    sc_ref, sh_ref = torch.zeros(G, N, dtype=torch.float64), torch.zeros(G, N, dtype=torch.float64)
    for g in range(G):
        vg = v[grp == g]
        s = gamma.double() / torch.sqrt(vg.var(0, unbiased=False) + 1e-5)
        sc_ref[g], sh_ref[g] = s, beta.double() - vg.mean(0) * s
    # the specification (float64 emulation) against the direct statistics
    Ge, Se = torch.zeros(cpu.T, K * K, dtype=torch.float64), torch.zeros(cpu.T, K, dtype=torch.float64)
    emu.gram_rows(X, K, sc, sh, cpu, Ge, Se)
    sce, she = torch.zeros(G, N), torch.zeros(G, N)
    emu.gn_finalize_gram_dbias(Ge, Se, cpu, tile_det, K, W, dbias, N, gamma, beta, 1e-5, None, sce, she)
    close(sce, sc_ref.float(), 2e-6, 'specification: scale')
    close(she, sh_ref.float(), 2e-6, 'specification: shift')
    # the device
    Gp = torch.zeros(gpu.T, K * K, dtype=torch.float64).cuda()
    Sp = torch.zeros(gpu.T, K, dtype=torch.float64).cuda()
    hip.gram_rows(X.cuda(), K, sc.cuda(), sh.cuda(), gpu, Gp, Sp)
    scg, shg = torch.full((G, N), float('nan')).cuda(), torch.full((G, N), float('nan')).cuda()
    work = torch.zeros(G, K * K + K, dtype=torch.float64).cuda()
    hip.gn_finalize_gram_dbias(Gp, Sp, gpu, tile_det.cuda(), K, W.cuda(), dbias.cuda(), N, gamma.cuda(), beta.cuda(), 1e-5,
                               work, scg, shg)
    close(scg, sc_ref.float(), 5e-6, 'scale from the Gram route with a per-detection bias')
    close(shg, sh_ref.float(), 5e-6, 'shift from the Gram route with a per-detection bias')


@pytest.mark.parametrize('N', [64, 128])
@pytest.mark.parametrize('counts', [[5, 300, 128], [1], [70001, 257]])
def test_pn_mlp64_matches_row_gemm_contract(hip, N, counts):
    """Persistent weight-resident K = 64 PointNet layer: raw output and per-tile statistics vs the fp64 emulation
    and vs mmmot_gemm_rows on the same hl16 weights; more tiles than resident workgroups ([70001, 257] -> 550 tiles
    on 512 slots) so the grid-stride loop and the register prefetch of a later tile are exercised."""
    from mmmot_amd.pack import hl16_weight_shift, to_hl16
    emu = TorchOps(torch.float64)
    t = DevTiles(counts)
    R, G_ = sum(counts), len(counts)
    X = rnd(R, 64, seed=90) + 0.3
    W = rnd(N, 64, seed=91, scale=0.125)
    bias = rnd(N, seed=92)
    sc, sh = rnd(G_, 64, seed=93).abs() + 0.5, rnd(G_, 64, seed=94)
    shift = hl16_weight_shift(W)
    W16 = to_hl16(W.double() * 2.0 ** shift)
    osv = 2.0 ** -shift
    Yr, pr = torch.zeros(R, N, dtype=torch.float64), torch.zeros(t.cpu.T, 2, N, dtype=torch.float64)
    emu.pn_mlp64(W16, osv, t.cpu, N, X, sc, sh, bias, Yr, pr)
    Yg, pg = torch.full((R, N), float('nan')).cuda(), torch.full((t.gpu.T, 2, N), float('nan')).cuda()
    hip.pn_mlp64(W16.cuda(), osv, t.gpu, N, X.cuda(), sc.cuda(), sh.cuda(), bias.cuda(), Yg, pg)
    close(Yg, Yr.float(), 2e-6, 'pn_mlp64 output')
    close(pg[:, 0], pr[:, 0].float(), 1e-5, 'pn_mlp64 tile sums')
    close(pg[:, 1], pr[:, 1].float(), 1e-4, 'pn_mlp64 tile M2')
    Y2, p2 = torch.full((R, N), float('nan')).cuda(), torch.full((t.gpu.T, 2, N), float('nan')).cuda()
    hip.gemm(W16.cuda(), t.gpu, N, 64, X=X.cuda(), bias=bias.cuda(), Y=Y2, part=p2, sc=sc.cuda(), sh=sh.cuda(),
             amode=1, w_hl16=True, oscale=osv)
    close(Yg, Y2, 1e-6, 'pn_mlp64 vs gemm_rows output')
    close(pg[:, 0], p2[:, 0], 1e-5, 'pn_mlp64 vs gemm_rows sums')


@pytest.mark.parametrize('C,C4,R', [(128, 64, 22), (256, 64, 5), (512, 128, 131), (512, 128, 1), (512, 128, 259),
                                    (128, 64, 2048), (256, 64, 301)])
def test_skippool_head_one_launch(hip, C, C4, R):
    """the fused SkipPool head (LayerNorm -> 1x1 -> LayerNorm + ReLU -> 1x1 -> LayerNorm + ReLU) vs its float64
    specification; partial last workgroup, strided output slice"""
    emu = TorchOps(torch.float64)
    hd = dict(g0=rnd(C, seed=1).abs() + 0.5, b0=rnd(C, seed=2), w1=rnd(C4, C, seed=3, scale=C ** -0.5), c1=rnd(C4, seed=4),
              g2=rnd(C4, seed=5).abs() + 0.5, b2=rnd(C4, seed=6), w4=rnd(128, C4, seed=7, scale=C4 ** -0.5),
              c4=rnd(128, seed=8), g5=rnd(128, seed=9).abs() + 0.5, b5=rnd(128, seed=10))
    P = rnd(R, C, seed=11) * 3 + 1
    ref = torch.zeros(R, 128, dtype=torch.float64)
    emu.skippool_head(P, C, hd, 1e-5, ref, R)
    cat = torch.full((R, 1024), 9.0).cuda()
    hip.skippool_head(P.cuda(), C, {k: v.cuda() for k, v in hd.items()}, 1e-5, cat[:, 256:384], R)
    close(cat[:, 256:384], ref.float(), 2e-5, 'skippool head')
    assert (cat[:, :256] == 9.0).all() and (cat[:, 384:] == 9.0).all()
    if R >= 256:
        # the batch form (8 rows per workgroup, the 64 partial sums of a step reduced by one transposing butterfly) against
        # the latency form (2 rows per workgroup, one butterfly per sum) on a slice of the same rows: bit for bit - a
        # detection's appearance feature does not depend on the batch it is evaluated in
        few = torch.zeros(37, 128).cuda()
        hip.skippool_head(P[100:137].cuda(), C, {k: v.cuda() for k, v in hd.items()}, 1e-5, few, 37)
        assert torch.equal(few, cat[100:137, 256:384])
