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
