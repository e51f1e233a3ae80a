"""The opt-in 'f16q8' trunk arithmetic (include/mmmot_hip.h "hq8"): fp16 main term on the fp16 matrix cores, the two
correction terms of the hi/lo split on the fp8 matrix cores (one block-scaled K=64 MFMA per product).

* record format: device encoder == host encoder (pack.to_hq8_act), bit for bit;
* the kernel computes exactly  hi*hi + 2^-3 (a8 * w_lo8 + a_lo8 * w8)  of the RECORDS it is given - checked with
  records whose fp16 parts are zero (only the fp8 path contributes) and whose fp8 parts are zero, so that a
  misplaced fp8 operand cannot hide behind the 2^-11 weight of the correction terms;
* geometry cases of the hl16 patch kernel, the fused conv1 head, the hq8 consumer (segment mean);
* end to end: every reference golden within the 1e-3 budget of BASELINE.json with trunk='f16q8'."""
import numpy as np
import pytest
import torch

from common import TOL, build_model, case_inputs, case_names, compare_outputs, get_case, golden
from fake_ops import TorchOps
from mmmot_amd.pack import (conv1_weight_shift, _e4m3, _hq8_records, from_hl16, from_hq8_act, hl16_weight_shift, hq8_parts, to_hl16,
                            to_hq8_act, to_hq8_w)
from mmmot_amd.plan import Segments
from test_conv_patch_gpu import CASES
from test_kernels_gpu import close, hip, rnd  # noqa: F401  (hip is a fixture)
from test_parity_gpu import to_dev

pytestmark = pytest.mark.gpu

# an hq8 OUTPUT is re-encoded: its decoded value carries the e4m3 rounding of the lo part, 2^-3 * 2^-11 relative;
# a rounding flip between kernel (fp32 accumulation) and emulation (fp64) moves a value by at most that much
ENC_TOL = 1.3e-4


def bytes_of(t):
    return t.detach().cpu().contiguous().view(torch.uint8)


def test_device_encoder_matches_host_encoder(hip):
    x = torch.cat([rnd(4096, 32, seed=1) * 3.0, rnd(64, 32, seed=2) * 300.0, rnd(64, 32, seed=3) * 1e-3,
                   torch.tensor([[0.0, -0.0, 1792.0, 1800.0, 7000.0, 65000.0, 7e4, -7e4] * 4])]).contiguous()
    y = torch.zeros_like(x).cuda()
    hip.hq8_pack(x.cuda(), y)
    want = to_hq8_act(x)
    assert torch.equal(bytes_of(y), bytes_of(want)), 'device hq8 encoder differs from pack.to_hq8_act'
    z = torch.zeros_like(x).cuda()
    hip.hq8_unpack(y, z)
    assert torch.equal(z.cpu(), from_hq8_act(want))
    # the decoded value is within 2^-14 of the input (values below the fp16 range limit)
    ok = x.abs() < 6e4
    assert ((z.cpu() - x).abs()[ok] <= x.abs()[ok] * 2.0 ** -14 + 2.0 ** -19).all()


def run_records(hip, xrec, wrec, bias, pool, L, H, W, Cin, Cout, oscale):
    emu = TorchOps(torch.float64)
    ref = emu._conv_hq8(hq8_parts(xrec), wrec, bias, L, H, W, Cin, Cout, bool(pool), oscale).float()
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
    hip.conv3x3_hq8(xrec.cuda(), wrec.cuda(), bias.cuda(), out, L, H, W, Cin, Cout, bool(pool),
                    oscale.cuda() if torch.is_tensor(oscale) else oscale)
    dec = torch.zeros_like(out)
    hip.hq8_unpack(out, dec)
    return dec, ref, out


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', CASES)
def test_conv3x3_hq8_geometry(hip, pool, L, H, W, Cin, Cout):
    x = torch.relu(rnd(L * H * W, Cin, seed=630)) * 3.0
    w = rnd(9, Cout, Cin, seed=631, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=632, scale=0.1)
    shift = hl16_weight_shift(w)
    dec, ref, _ = run_records(hip, to_hq8_act(x), to_hq8_w(w.double() * 2.0 ** shift), bias, pool, L, H, W, Cin, Cout,
                              2.0 ** -shift)
    close(dec, ref, ENC_TOL, 'hq8 conv vs fp64 statement of the hq8 arithmetic')
    # and the arithmetic is fp32-class: against the plain fp64 convolution of the decoded inputs / true weights
    emu = TorchOps(torch.float64)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    full = torch.zeros(L * Ho * Wo, Cout)
    emu.conv3x3(from_hq8_act(to_hq8_act(x)).view(L, H, W, Cin), from_hl16(to_hl16(w.double() * 2.0 ** shift)) * 2.0 ** -shift,
                bias, full, L, H, W, Cin, Cout, False, bool(pool))
    close(dec, full, 2.5e-4, 'hq8 conv vs exact convolution')


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', [(0, 7, 16, 16, 32, 128), (1, 12, 32, 32, 96, 64),
                                                 (0, 40, 16, 16, 128, 256), (1, 33, 24, 40, 64, 128), (1, 9, 4, 4, 512, 512)])
def test_conv3x3_hq8_chained_tiles(hip, pool, L, H, W, Cin, Cout):
    """several tiles per workgroup (persistent grid capped at 8): chained tiles give the bytes of the unchained launch"""
    from mmmot_amd import _lib
    lib = _lib.load()
    x = torch.relu(rnd(L * H * W, Cin, seed=680)) * 3.0
    w = rnd(9, Cout, Cin, seed=681, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=682, scale=0.1).cuda()
    shift = hl16_weight_shift(w)
    xs, ws = to_hq8_act(x).cuda(), to_hq8_w(w.double() * 2.0 ** shift).cuda()
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    outs = []
    try:
        for limit in (0, 8):
            assert lib.mmmot_set_patch_grid_limit(limit) == 0
            o = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
            hip.conv3x3_hq8(xs, ws, bias, o, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
            outs.append(bytes_of(o))
    finally:
        lib.mmmot_set_patch_grid_limit(0)
    assert torch.equal(outs[0], outs[1]), 'chained and unchained hq8 launches differ'
    dec = torch.zeros(L * Ho * Wo, Cout).cuda()
    hip.hq8_unpack(outs[1].cuda().view(torch.float32).view(L * Ho * Wo, Cout), dec)
    emu = TorchOps(torch.float64)
    ref = emu._conv_hq8(hq8_parts(to_hq8_act(x)), ws.cpu(), bias.cpu(), L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift).float()
    close(dec, ref, ENC_TOL, 'chained hq8 tiles vs fp64 statement')


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', [(0, 40, 4, 4, 64, 128), (1, 37, 4, 4, 96, 64), (0, 5, 3, 3, 64, 64),
                                                 (1, 18, 2, 2, 128, 128), (1, 70, 4, 4, 512, 512)])
def test_conv3x3_hq8_whole_map_blocks(hip, pool, L, H, W, Cin, Cout):
    """maps of at most 4 x 4 pixels in the whole-map geometry (16 maps per tile, no halo): the fp64 statement of the hq8
    arithmetic, and the bytes of the haloed 8 x 8 geometry and of the chained launch"""
    from mmmot_amd import _lib
    lib = _lib.load()
    x = torch.relu(rnd(L * H * W, Cin, seed=690)) * 3.0
    w = rnd(9, Cout, Cin, seed=691, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=692, scale=0.1)
    shift = hl16_weight_shift(w)
    xrec, wrec = to_hq8_act(x), to_hq8_w(w.double() * 2.0 ** shift)
    dec, ref, out = run_records(hip, xrec, wrec, bias, pool, L, H, W, Cin, Cout, 2.0 ** -shift)
    close(dec, ref, ENC_TOL, 'hq8 whole-map blocks vs fp64 statement of the hq8 arithmetic')
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    xs, ws, bs = xrec.cuda(), wrec.cuda(), bias.cuda()
    for setter, arg, what in ((lib.mmmot_set_patch_min_block, 8, 'haloed 8x8 geometry'),
                              (lib.mmmot_set_patch_grid_limit, 8, 'chained launch')):
        assert setter(arg) == 0
        try:
            o = torch.full((L * Ho * Wo, Cout), float('nan')).cuda()
            hip.conv3x3_hq8(xs, ws, bs, o, L, H, W, Cin, Cout, bool(pool), 2.0 ** -shift)
        finally:
            assert setter(0) == 0
        assert torch.equal(bytes_of(out), bytes_of(o)), 'hq8 whole-map blocks differ from the ' + what


@pytest.mark.parametrize('which', ['fp8_only', 'fp16_only', 'a8_only', 'al8_only'])
@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', [(0, 2, 16, 16, 64, 128), (1, 3, 8, 8, 96, 64)])
def test_conv3x3_hq8_operand_placement(hip, which, pool, L, H, W, Cin, Cout):
    """records assembled from INDEPENDENT random parts: each operand class must land in its own k-slots"""
    g = torch.Generator().manual_seed(640)
    R = L * H * W
    ah = torch.rand(R, Cin, generator=g, dtype=torch.float64).to(torch.float16)
    a8 = _e4m3(torch.rand(R, Cin, generator=g, dtype=torch.float64) * 4.0)
    al8 = _e4m3((torch.rand(R, Cin, generator=g, dtype=torch.float64) - 0.5) * 4.0)
    wh = ((torch.rand(9 * Cout, Cin, generator=g, dtype=torch.float64) - 0.5) * 0.25).to(torch.float16)
    wl8 = _e4m3((torch.rand(9 * Cout, Cin, generator=g, dtype=torch.float64) - 0.5) * 2.0)
    w8 = _e4m3((torch.rand(9 * Cout, Cin, generator=g, dtype=torch.float64) - 0.5) * 2.0)
    z8a, z8w = torch.zeros_like(a8), torch.zeros_like(wl8)
    if which == 'fp8_only':
        ah, wh = torch.zeros_like(ah), torch.zeros_like(wh)
    elif which == 'fp16_only':
        a8, al8, wl8, w8 = z8a, z8a, z8w, z8w
    elif which == 'a8_only':      # a8 pairs with w_lo8 only: al8 * w8 removed, and w8 left in place must not leak
        ah, wh, al8 = torch.zeros_like(ah), torch.zeros_like(wh), z8a
    else:                         # al8 pairs with w8 only
        ah, wh, a8 = torch.zeros_like(ah), torch.zeros_like(wh), z8a
    xrec, wrec = _hq8_records(ah, a8, al8), _hq8_records(wh, wl8, w8)
    bias = rnd(Cout, seed=641, scale=0.1)
    dec, ref, _ = run_records(hip, xrec, wrec.view(9, Cout, Cin), bias, pool, L, H, W, Cin, Cout, 0.5)
    assert float(ref.abs().max()) > 1.0
    close(dec, ref, ENC_TOL, 'hq8 operand placement (%s)' % which)


def test_hq8_is_deterministic(hip):
    pool, L, H, W, Cin, Cout = 0, 6, 16, 16, 256, 256
    x = torch.relu(rnd(L * H * W, Cin, seed=650)) * 3.0
    w = rnd(9, Cout, Cin, seed=651, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=652, scale=0.1).cuda()
    shift = hl16_weight_shift(w)
    xs, ws = to_hq8_act(x).cuda(), to_hq8_w(w.double() * 2.0 ** shift).cuda()
    first = None
    for _ in range(20):
        o = torch.zeros(L * H * W, Cout).cuda()
        hip.conv3x3_hq8(xs, ws, bias, o, L, H, W, Cin, Cout, False, 2.0 ** -shift)
        if first is None:
            first = o.clone()
        else:
            assert torch.equal(bytes_of(first), bytes_of(o)), 'hq8 patch kernel is not deterministic across launches'


@pytest.mark.parametrize('L,H,W', [(2, 16, 16), (3, 32, 48), (1, 14, 22), (5, 64, 64), (2, 8, 8)])
def test_conv1_fused_hq8(hip, L, H, W):
    crops = rnd(L, 3, H, W, seed=660) * 1.5
    w1 = rnd(64, 3, 3, 3, seed=661, scale=(2.0 / 27) ** 0.5)
    b1 = rnd(64, seed=662, scale=0.1)
    w2 = rnd(9, 64, 64, seed=663, scale=(2.0 / 576) ** 0.5)
    b2 = rnd(64, seed=664, scale=0.1)
    w1p = torch.zeros(64, 32)
    w1p[:, :27] = w1.permute(0, 2, 3, 1).reshape(64, 27)
    s1, s2 = conv1_weight_shift(w1p, b1), hl16_weight_shift(w2)
    w1h, w2q = to_hl16(w1p.double() * 2.0 ** s1), to_hq8_w(w2.double() * 2.0 ** s2)
    emu = TorchOps(torch.float64)
    want = torch.zeros(L * (H // 2) * (W // 2), 64)
    emu.conv1_fused_hq8(crops, w1h, b1, 2.0 ** -s1, w2q, b2, 2.0 ** -s2, want, L, H, W)
    out = torch.full((L * (H // 2) * (W // 2), 64), float('nan')).cuda()
    hip.conv1_fused_hq8(crops.cuda(), w1h.cuda(), b1.cuda(), 2.0 ** -s1, w2q.cuda(), b2.cuda(), 2.0 ** -s2, out, L, H, W)
    dec = torch.zeros_like(out)
    hip.hq8_unpack(out, dec)
    # conv1_1 runs in fp32 on the device and in fp64 in the emulation: its re-encoding may flip an e4m3 rounding
    close(dec, from_hq8_act(want), 2.5e-4, 'fused conv1 (hq8) vs emulation')


def test_segment_mean_reads_hq8(hip):
    rows, C = 37 * 9, 128
    x = torch.relu(rnd(rows, C, seed=670)) * 2.0
    rec = to_hq8_act(x)
    segs = Segments(np.arange(0, rows, 9), np.full(37, 9), np.ones(37), np.zeros(37), 'cuda')
    out = torch.zeros(37, C).cuda()
    hip.segment_mean(rec.cuda(), C, segs, out, use_group=False, hl16=2)
    want = from_hq8_act(rec).double().view(37, 9, C).mean(1).float()
    close(out, want, 2e-6, 'segment mean over hq8 rows')


@pytest.mark.parametrize('pool,L,H,W,Cin,Cout', [(0, 3, 16, 16, 64, 128), (1, 2, 8, 8, 128, 64), (1, 2, 32, 32, 64, 256)])
def test_conv3x3_hq8_per_channel_scales(hip, pool, L, H, W, Cin, Cout):
    """output channels with gains spread over 1e6, every channel with its own power-of-two weight scale and the
    [Cout] vector undoing it in the epilogue: every channel matches the emulation of the hq8 arithmetic (what the
    per-channel scales buy in accuracy is pinned on the host side, tests/test_hq8_pack_cpu.py)"""
    from mmmot_amd.pack import hl16_channel_shifts
    g = torch.Generator().manual_seed(78)
    gain = torch.pow(10.0, torch.rand(Cout, generator=g) * 6.0 - 4.0).double()
    x = torch.relu(rnd(L * H * W, Cin, seed=490)) * 3.0
    w = rnd(9, Cout, Cin, seed=491, scale=(2.0 / (9 * Cin)) ** 0.5).double() * gain.view(1, -1, 1)
    bias = (rnd(Cout, seed=492, scale=0.1).double() * gain).float()
    shifts = hl16_channel_shifts(w)
    osc = torch.pow(2.0, -shifts.double()).float()
    xrec, wrec = to_hq8_act(x), to_hq8_w(w * torch.pow(2.0, shifts.double()).view(1, -1, 1))
    dec, ref, _ = run_records(hip, xrec, wrec, bias, pool, L, H, W, Cin, Cout, osc)
    # per channel: the encoding tolerance relative to the channel's largest output, plus the absolute quantum of the
    # hq8 OUTPUT record (e4m3(512 lo) is subnormal below lo = 2^-15: 2^-18 absolute) - the low-gain channels' outputs
    # are ~1e-3 here
    err, chmax = (dec.cpu().double() - ref.double()).abs().amax(dim=0), ref.double().abs().amax(dim=0)
    print('hq8 per-channel scales: worst per-channel relative deviation from the emulation %.2e (output maxima %.1e .. %.1e)'
          % ((err / chmax.clamp_min(1e-30)).max().item(), chmax.min().item(), chmax.max().item()))
    assert (err <= ENC_TOL * chmax + 2.0 ** -17).all(), 'kernel differs from the emulation of the hq8 arithmetic'


@pytest.mark.parametrize('name', case_names())
def test_f16q8_forward_matches_reference_golden(name):
    c, base = get_case(name)
    m = build_model(c, base, device='cuda')
    m.set_trunk('f16q8')
    m.engine().q8_min_crop = 0  # the fixtures include 32-pixel crops: exercise the e4m3 arithmetic on them too
    m.engine().q8_layers = None  # ... on every trunk layer (the default keeps the first three in f16x3)
    assert m.engine().ops.name == 'hip' and m.engine().trunk == 'f16q8'
    with torch.no_grad():
        out = m(*to_dev(case_inputs(c)))
    errs = compare_outputs(out, golden(name), tol=TOL)
    print(name, {k: '%.1e' % v for k, v in errs.items()})


def test_f16q8_unfused_first_layer_matches_golden(monkeypatch):
    monkeypatch.setenv('MMMOT_FUSE_CONV1', '0')
    name = 's2_C_minus_abs_dual_add'
    c, base = get_case(name)
    m = build_model(c, base, device='cuda')
    m.set_trunk('f16q8')
    m.engine().q8_min_crop = 0
    m.engine().q8_layers = None
    with torch.no_grad():
        out = m(*to_dev(case_inputs(c)))
    compare_outputs(out, golden(name), tol=TOL)


def test_conv1_fused_hq8_many_tiles_per_workgroup(hip):
    from mmmot_amd import _lib
    lib = _lib.load()
    assert lib.mmmot_set_patch_grid_limit(8) == 0
    try:
        test_conv1_fused_hq8(hip, 5, 64, 64)
        test_conv1_fused_hq8(hip, 3, 32, 48)
    finally:
        lib.mmmot_set_patch_grid_limit(0)


SURVEY = [  # (fusion, affinity, softmax, N, M, S, pts)
    ('A', 'multiply', 'none', 4, 6, 224, 100), ('B', 'multiply', 'none', 10, 12, 128, 200),
    ('C', 'multiply', 'none', 16, 16, 64, 128), ('C', 'minus_abs', 'dual_add', 2, 30, 96, 40),
    ('C', 'minus_abs', 'dual_add', 25, 3, 160, 64), ('A', 'minus_abs', 'single', 7, 7, 32, 300),
    ('B', 'multiply', 'dual', 12, 5, 192, 16), ('C', 'multiply', 'dual_max', 1, 1, 224, 2000),
    ('C', 'multiply', 'none', 32, 32, 32, 64), ('A', 'multiply', 'none', 3, 3, 256, 500),
]


@pytest.mark.parametrize('fusion,aff,sm,N,M,S,pts', SURVEY)
def test_f16q8_fresh_inputs_against_oracle(fusion, aff, sm, N, M, S, pts):
    """the opt-in f16q8 trunk arithmetic carries e4m3 correction terms: beyond the golden fixtures, fresh seeds / crop
    sizes / ragged point counts against the CPU oracle, every output within the 1e-3 budget (margins are printed)"""
    from mmmot_amd.synth import make_pair
    from oracle import restatement as R
    c, base = get_case('s2_C_minus_abs_dual_add')
    c = dict(c, fusion=fusion, aff=aff, sm=sm)
    m = build_model(c, base)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to('cuda')
    m.set_trunk('f16q8')
    m.engine().q8_min_crop = 0  # force the e4m3 arithmetic below the default minimum crop side as well
    m.engine().q8_layers = None  # and on every trunk layer
    cfg = dict(fusion=fusion, affinity_op=aff, softmax_mode=sm, neg_threshold=base['neg_threshold'],
               score_arch=base['score_arch'])
    worst = {}
    for seed in (0, 1):
        dets, info, ds = make_pair(N, M, S, pts, seed=900 + 7 * seed + N + M, ragged=True)
        with torch.no_grad():
            ref = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], [N, M])
            out = m(*to_dev((dets, info, ds)))
        for a, b, what in ((out[0], ref[0], 'det'), (out[1][0], ref[1][0], 'link'), (out[2], ref[2], 'new'),
                           (out[3], ref[3], 'end')):
            err = (a.cpu() - b).abs().max().item()
            worst[what] = max(worst.get(what, 0.0), err)
            assert err < TOL, (seed, what, err)
    print('f16q8 %s/%s/%s N=%d M=%d S=%d pts=%d: %s' % (fusion, aff, sm, N, M, S, pts,
                                                      {k: '%.1e' % v for k, v in worst.items()}))


def test_f16q8_small_crops_use_the_f16x3_trunk():
    """default policy: crops below 64 pixels run the fp32-class trunk (f16q8 would cost up to 6e-4 there)"""
    name = 's1_C_multiply_none'
    c, base = get_case(name)
    assert c['S'] < 64
    outs = []
    for trunk in ('f16q8', 'f16x3'):
        m = build_model(c, base, device='cuda')
        m.set_trunk(trunk)
        assert m.engine().q8_min_crop == 64
        with torch.no_grad():
            outs.append(m(*to_dev(case_inputs(c))))
    for a, b in zip(outs[0][:1] + tuple(outs[0][1]) + outs[0][2:4], outs[1][:1] + tuple(outs[1][1]) + outs[1][2:4]):
        assert torch.equal(a, b), 'f16q8 below the minimum crop side must be the f16x3 path'
    compare_outputs(outs[0], golden(name), tol=TOL)
