"""mmmot_gemm_rows with w_hl16=1: the row GEMM on the fp16 matrix cores (3-term hi/lo split) must agree
with an fp64 GEMM to fp32-class accuracy in every A-operand mode."""
import pytest
import torch

from fake_ops import TorchOps
from mmmot_amd.pack import hl16_weight_shift, to_hl16
from test_kernels_gpu import DevTiles, close, hip, rnd  # noqa: F401  (hip is a fixture)

pytestmark = pytest.mark.gpu


def split_w(W):
    shift = hl16_weight_shift(W)
    return to_hl16(W.double() * 2.0 ** shift), 2.0 ** -shift


@pytest.mark.parametrize('N,K', [(64, 64), (128, 64), (512, 512), (1024, 128), (256, 512), (64, 1024)])
@pytest.mark.parametrize('counts', [[5, 300, 128], [1]])
def test_gemm_f16_plain_stats(hip, N, K, counts):
    emu = TorchOps(torch.float64)
    tl = DevTiles(counts)
    R = sum(counts)
    X = rnd(R, K, seed=210) + 3.0
    W = rnd(N, K, seed=211, scale=K ** -0.5)
    bias = rnd(N, seed=212, scale=0.2)
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    emu.gemm(W, tl.cpu, N, K, X=X, bias=bias, Y=Y, part=part, act=1)
    W16, osc = split_w(W)
    Yg, pg = torch.full((R, N), float('nan')).cuda(), torch.full((tl.cpu.T, 2, N), float('nan')).cuda()
    hip.gemm(W16.cuda(), tl.gpu, N, K, X=X.cuda(), bias=bias.cuda(), Y=Yg, part=pg, act=1, w_hl16=True, oscale=osc)
    close(Yg, Y, 3e-6, 'gemm f16x3 Y')
    close(pg[:, 0], part[:, 0], 1e-5, 'gemm f16x3 tile sums')
    close(pg[:, 1], part[:, 1], 1e-4, 'gemm f16x3 tile M2')


def test_gemm_f16_norm_relu_dbias_strided(hip):
    emu = TorchOps(torch.float64)
    counts = [70, 200]
    tl = DevTiles(counts)
    N, K, R = 128, 64, 270
    buf = rnd(R, 2 * K, seed=220)
    W = rnd(N, K, seed=221, scale=K ** -0.5)
    sc, sh = rnd(2, K, seed=222).abs() + 0.5, rnd(2, K, seed=223)
    dbias = rnd(9, N, seed=224)
    rowidx = (torch.arange(R) * 9 // R).int()
    out = torch.zeros(R, 3 * N)
    emu.gemm(W, tl.cpu, N, K, X=buf[:, K:], Y=out[:, N:2 * N], sc=sc, sh=sh, amode=1, dbias=dbias, rowidx=rowidx)
    W16, osc = split_w(W)
    outg, bufg = torch.zeros(R, 3 * N).cuda(), buf.cuda()
    hip.gemm(W16.cuda(), tl.gpu, N, K, X=bufg[:, K:], Y=outg[:, N:2 * N], sc=sc.cuda(), sh=sh.cuda(), amode=1,
             dbias=dbias.cuda(), rowidx=rowidx.cuda(), w_hl16=True, oscale=osc)
    close(outg, out, 3e-6, 'gemm f16x3 norm_relu + dbias')


@pytest.mark.parametrize('pairop', [0, 1, 2])
def test_gemm_f16_pair(hip, pairop):
    emu = TorchOps(torch.float64)
    NM = [(5, 7), (130, 3)]
    counts = [n * m for n, m in NM]
    tl = DevTiles(counts)
    K, N = 512, 1024
    Fm = rnd(150, K, seed=230)
    W = rnd(N, K, seed=231, scale=K ** -0.5)
    bias = rnd(N, seed=232)
    mk = lambda dev: dict(row0=torch.tensor(tl.cpu.h_g_row0).to(dev), M=torch.tensor([7, 3], dtype=torch.int32).to(dev),
                          aoff=torch.tensor([0, 12], dtype=torch.int32).to(dev),
                          boff=torch.tensor([5, 142], dtype=torch.int32).to(dev))
    R = sum(counts)
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    emu.gemm(W, tl.cpu, N, K, FA=Fm, FB=Fm, pair=mk('cpu'), amode=2, pairop=pairop, bias=bias, Y=Y, part=part)
    W16, osc = split_w(W)
    Yg, pg = torch.zeros(R, N).cuda(), torch.zeros(tl.cpu.T, 2, N).cuda()
    Fg = Fm.cuda()
    hip.gemm(W16.cuda(), tl.gpu, N, K, FA=Fg, FB=Fg, pair=mk('cuda'), amode=2, pairop=pairop, bias=bias.cuda(), Y=Yg,
             part=pg, w_hl16=True, oscale=osc)
    close(Yg, Y, 3e-6, 'gemm f16x3 pair')
    close(pg[:, 0], part[:, 0], 1e-5, 'pair f16x3 tile sums')


# ---- the wide kernel (csrc/gemm_wide.hip): A fragments generated in registers, 32 rows x 256 columns per wave ----
@pytest.fixture
def variants(hip):
    from mmmot_amd import _lib
    lib = _lib.load()
    yield lib.mmmot_set_gemm_rows_variant
    assert lib.mmmot_set_gemm_rows_variant(0) == 0


def _pair_tables(tl, NM, aoff, boff, dev):
    return dict(row0=torch.tensor(tl.cpu.h_g_row0).to(dev), M=torch.tensor([m for _, m in NM], dtype=torch.int32).to(dev),
                aoff=torch.tensor(aoff, dtype=torch.int32).to(dev), boff=torch.tensor(boff, dtype=torch.int32).to(dev),
                uniform32=all(m % 32 == 0 for _, m in NM))


@pytest.mark.parametrize('pairop', [0, 1, 2])
@pytest.mark.parametrize('N,K,NM', [(1024, 512, [(5, 64), (130, 32), (64, 64)]), (512, 512, [(20, 96)]), (128, 512, [(9, 32)]),
                                    (512, 256, [(40, 32), (1, 32)]), (1024, 512, [(128, 128), (3, 160)]),
                                    (1024, 512, [(5, 7), (130, 3)])])   # last: M % 32 != 0 - the tile kernel serves it
def test_gemm_wide_pair(hip, variants, pairop, N, K, NM):
    """stacked pairwise layer (reference modules/gcn.py:59-82, new_end.py:48-52) on the wide kernel: fp64 statement, and
    the tile kernel's Y and statistics bit for bit (same operand values, same accumulation and summation order)"""
    emu = TorchOps(torch.float64)
    counts = [n * m for n, m in NM]
    tl = DevTiles(counts)
    nf = sum(n + m for n, m in NM)
    Fm = rnd(nf + 3, 512, seed=240)
    aoff, boff, o = [], [], 1
    for n, m in NM:
        aoff.append(o)
        boff.append(o + n)
        o += n + m
    W = rnd(N, K, seed=241, scale=K ** -0.5)
    bias = rnd(N, seed=242)
    R = sum(counts)
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    cp = _pair_tables(tl, NM, aoff, boff, 'cpu')
    emu.gemm(W, tl.cpu, N, K, FA=Fm, FB=Fm, pair=cp, amode=2, pairop=pairop, bias=bias, Y=Y, part=part)
    W16, osc = split_w(W)
    Fg, Wg, bg, pt = Fm.cuda(), W16.cuda(), bias.cuda(), _pair_tables(tl, NM, aoff, boff, 'cuda')
    outs = {}
    for v in (3, 4, 1):
        assert variants(v) == 0
        Yg, pg = torch.full((R + 1, N), float('nan')).cuda(), torch.full((tl.cpu.T, 2, N), float('nan')).cuda()
        hip.gemm(Wg, tl.gpu, N, K, FA=Fg, FB=Fg, pair=pt, amode=2, pairop=pairop, bias=bg, Y=Yg, part=pg, w_hl16=True,
                 oscale=osc)
        assert torch.isnan(Yg[R]).all(), 'row beyond the last tile written'
        outs[v] = (Yg[:R].cpu(), pg.cpu())
    close(outs[3][0], Y, 3e-6, 'wide gemm pair Y')
    close(outs[3][1][:, 0], part[:, 0], 1e-5, 'wide gemm pair tile sums')
    close(outs[3][1][:, 1], part[:, 1], 1e-4, 'wide gemm pair tile M2')
    # NH = 1 (variant 3) and NH = 2 (variant 4); layers the wide kernel does not take fall through to the tile kernel
    for v in (3, 4):
        assert torch.equal(outs[v][0], outs[1][0]) and torch.equal(outs[v][1], outs[1][1]), 'wide kernel (variant %d) differs from the tile kernel' % v


@pytest.mark.parametrize('N,K,ldx,counts', [(512, 512, 1024, [300, 5, 128, 1000]), (128, 512, 512, [77, 260]),
                                            (1024, 256, 256, [129]), (256, 512, 512, [128] * 37)])
def test_gemm_wide_norm_relu(hip, variants, N, K, ldx, counts):
    """the GroupNorm-fed layers behind it (w_link.conv1.3 / conv1.6, reference modules/gcn.py:61-65): strided input rows
    (the conv1.0 half of the stacked layer's output), per-group scale / shift"""
    emu = TorchOps(torch.float64)
    tl = DevTiles(counts)
    R, G = sum(counts), len(counts)
    buf = rnd(R, ldx, seed=250)
    X = buf[:, ldx - K:]
    W = rnd(N, K, seed=251, scale=K ** -0.5)
    bias = rnd(N, seed=252, scale=0.2)
    sc, sh = rnd(G, K, seed=253).abs() + 0.5, rnd(G, K, seed=254)
    Y, part = torch.zeros(R, N), torch.zeros(tl.cpu.T, 2, N)
    emu.gemm(W, tl.cpu, N, K, X=X, bias=bias, Y=Y, part=part, sc=sc, sh=sh, amode=1)
    W16, osc = split_w(W)
    bufg = buf.cuda()
    outs = {}
    for v in (3, 4, 1):
        assert variants(v) == 0
        Yg, pg = torch.full((R, N), float('nan')).cuda(), torch.full((tl.cpu.T, 2, N), float('nan')).cuda()
        hip.gemm(W16.cuda(), tl.gpu, N, K, X=bufg[:, ldx - K:], bias=bias.cuda(), Y=Yg, part=pg, sc=sc.cuda(), sh=sh.cuda(),
                 amode=1, w_hl16=True, oscale=osc)
        outs[v] = (Yg.cpu(), pg.cpu())
    close(outs[3][0], Y, 3e-6, 'wide gemm norm_relu Y')
    close(outs[3][1][:, 0], part[:, 0], 1e-5, 'wide gemm norm_relu tile sums')
    close(outs[3][1][:, 1], part[:, 1], 1e-4, 'wide gemm norm_relu tile M2')
    for v in (3, 4):
        assert torch.equal(outs[v][0], outs[1][0]) and torch.equal(outs[v][1], outs[1][1]), 'wide kernel (variant %d) differs from the tile kernel' % v


def test_gemm_wide_long_chains_are_deterministic(hip, variants):
    """more items than workgroups (each persistent workgroup walks a chain of tiles, the successor's first weight stage and
    source rows requested during the last stage): 40 000 pair rows x 1024 columns = 626 items on 256 CUs, repeated"""
    NM = [(250, 160)]
    tl = DevTiles([40000])
    N, K = 1024, 512
    Fm = rnd(410, K, seed=260).cuda()
    W = rnd(N, K, seed=261, scale=K ** -0.5)
    W16, osc = split_w(W)
    pt = _pair_tables(tl, NM, [0], [250], 'cuda')
    outs = []
    for v in (1, 3, 3, 3, 4, 4, 4, 0):
        assert variants(v) == 0
        Yg, pg = torch.full((40000, N), float('nan')).cuda(), torch.full((tl.cpu.T, 2, N), float('nan')).cuda()
        hip.gemm(W16.cuda(), tl.gpu, N, K, FA=Fm, FB=Fm, pair=pt, amode=2, pairop=0, Y=Yg, part=pg, w_hl16=True, oscale=osc)
        outs.append((Yg, pg))
    for Yg, pg in outs[1:]:
        assert torch.equal(Yg, outs[0][0]), 'wide kernel (chained tiles) differs from the tile kernel'
    assert all(torch.equal(outs[1][1], outs[i][1]) for i in (2, 3, 4, 5, 6, 7))
    close(outs[1][1][:, 0].cpu(), outs[0][1][:, 0].cpu(), 1e-5, 'tile sums wide vs tile kernel')
