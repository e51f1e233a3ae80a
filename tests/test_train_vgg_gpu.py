"""Training-mode image encoder on the device (csrc/train_vgg.hip, conv3x3.hip RAW mode, csrc/backward.hip through the
C-ABI): every new kernel against its executable specification (tests/fake_ops.py, float64), the features and gradients
of ``appearance_autograd`` against torch.autograd through the oracle's training-mode trunk, and one SGD step on the WHOLE
network (tracking_model.py:50-66) against the oracle's."""
import pytest
import torch

from common import build_model, case_inputs, get_case
from fake_ops import TorchOps
from mmmot_amd import TrackingLoss
from mmmot_amd.plan import RowTiles
from mmmot_amd.train_vgg import appearance_autograd
from oracle import restatement as R
from test_kernels_gpu import close, hip, rnd  # noqa: F401  (hip is a fixture)
from test_train_cpu import make_gts
from test_train_vgg_cpu import oracle_leaves, sgd_step_reference_full

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('first,L,H,W,Cin,Cout', [(1, 3, 10, 12, 3, 64), (0, 2, 7, 9, 64, 128), (0, 3, 16, 16, 128, 64),
                                                    (0, 1, 4, 4, 512, 512)])
def test_conv3x3_raw(hip, first, L, H, W, Cin, Cout):
    emu = TorchOps(torch.float64)
    x = rnd(L, 3, H, W, seed=3) if first else rnd(L * H * W, Cin, seed=3)
    wp = rnd(Cout, 32, seed=4, scale=0.2) if first else rnd(9, Cout, Cin, seed=4, scale=(2.0 / (9 * Cin)) ** 0.5)
    if first:
        wp[:, 27:] = 0
    bias = rnd(Cout, seed=5, scale=0.1)
    ref = torch.zeros(L * H * W, Cout, dtype=torch.float64)
    emu.conv3x3_raw(x, wp, bias, ref, L, H, W, Cin, Cout, first)
    out = torch.full((L * H * W, Cout), float('nan')).cuda()
    hip.conv3x3_raw(x.cuda(), wp.cuda(), bias.cuda(), out, L, H, W, Cin, Cout, first)
    close(out, ref.float(), 1e-5, 'conv3x3 raw')  # K up to 4608 in a sequential fp32 chain
    assert (out < 0).any()  # no ReLU


@pytest.mark.parametrize('L,H,W,Cin,Cout,scale', [(3, 10, 12, 64, 128, 1.0), (2, 7, 9, 128, 64, 3e-6), (5, 4, 4, 512, 512, 1e-4),
                                                  (2, 16, 16, 256, 256, 1.0), (37, 4, 4, 128, 128, 2e-7), (1, 30, 20, 64, 64, 1.0)])
def test_conv3x3_raw_hl16_with_device_side_scales(hip, L, H, W, Cin, Cout, scale):
    """the training-mode trunk convolution on the fp16 matrix cores (forward: activations / dgrad: a gradient of magnitude
    `scale`): input and weights scaled by device-side powers of two, split, convolved by the trunk kernel with raw fp32
    output, unscaled by the per-channel vector - fp32-class RELATIVE accuracy against the fp64 convolution, also for
    gradients far below fp16's normal range"""
    emu = TorchOps(torch.float64)
    x = rnd(L * H * W, Cin, seed=30) * scale
    wp = rnd(9, Cout, Cin, seed=31, scale=(2.0 / (9 * Cin)) ** 0.5)
    bias = rnd(Cout, seed=32, scale=0.1) * scale
    ref = torch.zeros(L * H * W, Cout, dtype=torch.float64)
    emu.conv3x3_raw(x, wp, bias, ref, L, H, W, Cin, Cout, False)
    xg, wg = x.cuda(), wp.cuda()
    new = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=DEV)
    amx, amw, x16, w16, osc = new(1), new(1), new(L * H * W, Cin), new(9, Cout, Cin), new(Cout)
    hip.absmax(xg, amx)
    hip.absmax(wg, amw)
    assert abs(amx.item() - x.abs().max().item()) == 0.0 and abs(amw.item() - wp.abs().max().item()) == 0.0
    hip.hl16_pack_pow2(xg, x16, amx, 11)
    hip.hl16_pack_pow2(wg, w16, amw, 14)
    hip.pow2_oscale(osc, amx, 11, amw, 14)
    out = torch.full((L * H * W, Cout), float('nan')).cuda()
    hip.conv3x3_raw_hl16(x16, w16, bias.cuda(), out, L, H, W, Cin, Cout, osc)
    close(out / scale, ref.float() / scale, 3e-6, 'conv3x3 raw hl16 (device-side power-of-two scales)')
    assert (out < 0).any()  # no ReLU


@pytest.mark.parametrize('amax', [0.0, 1e-45, 1e-40, float('inf'), float('nan'), 3e38])
def test_device_side_scales_survive_degenerate_maxima(hip, amax):
    """a zero / denormal / non-finite maximum must not become a 2^(+-huge) scale (ldexpf -> inf, 0 * inf = NaN
    in the products): the exponent is clamped to +-100 and a non-finite or non-positive maximum means 'unscaled' """
    x = rnd(64, 64, seed=40).cuda() * (0.0 if amax == 0.0 else 1.0)
    am = torch.tensor([amax], dtype=torch.float32, device=DEV)
    y16, osc = torch.empty(64, 64, device=DEV), torch.empty(64, device=DEV)
    hip.hl16_pack_pow2(x, y16, am, 11)
    hip.pow2_oscale(osc, am, 11, am, 14)
    from mmmot_amd.pack import from_hl16
    back = from_hl16(y16.cpu())
    # (every exponent is clamped to +-100 and the sum of two to +-126: no 2^(+-huge) = inf, no 0 * inf = NaN)
    assert torch.isfinite(back).all() and torch.isfinite(osc).all() and (osc > 0).all()
    # whatever scale was chosen, unscaling by the vector's factor for ONE operand gives the input back (fp16-split accuracy)
    one = torch.empty(64, device=DEV)
    hip.pow2_oscale(one, am, 11, None, 0)
    if not (0.0 < amax < float('inf')):  # "unscaled": the split of the data itself (a finite maximum that lies about the
        assert one[0].item() == 1.0       # O(1) data - the other cases - only has to stay finite)
        close(back.cuda(), x, 1e-5, 'pack_pow2 round trip')


def test_rows_stats_and_bn_relu_pool(hip):
    emu = TorchOps(torch.float64)
    L, H, W, C = 3, 9, 7, 64
    Z = rnd(L * H * W, C, seed=6) * 2 + 0.3
    for counts in ([L * H * W], [1] * 11):
        cpu, gpu = RowTiles(counts, 'cpu'), RowTiles(counts, DEV)
        pr = torch.zeros(cpu.T, 2, C, dtype=torch.float64)
        emu.rows_stats(Z, C, cpu, pr)
        pg = torch.full((gpu.T, 2, C), float('nan')).cuda()
        hip.rows_stats(Z.cuda(), C, gpu, pg)
        close(pg, pr.float(), 2e-6, 'rows_stats')
    sc, sh = rnd(C, seed=7).abs() + 0.5, rnd(C, seed=8)
    for pool in (0, 1):
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        ar = torch.zeros(L * Ho * Wo, C, dtype=torch.float64)
        emu.bn_relu_pool(Z, C, sc, sh, L, H, W, pool, ar)
        ag = torch.full((L * Ho * Wo, C), float('nan')).cuda()
        hip.bn_relu_pool(Z.cuda(), C, sc.cuda(), sh.cuda(), L, H, W, pool, ag)
        close(ag, ar.float(), 2e-6, 'bn_relu_pool pool=%d' % pool)
    dP = rnd(L * (H // 2) * (W // 2), C, seed=9)
    dr = torch.zeros(L * H * W, C, dtype=torch.float64)
    emu.maxpool_bwd(Z, C, sc, sh, dP, L, H, W, dr)
    dg = torch.full((L * H * W, C), float('nan')).cuda()
    hip.maxpool_bwd(Z.cuda(), C, sc.cuda(), sh.cuda(), dP.cuda(), L, H, W, dg)
    close(dg, dr.float(), 1e-6, 'maxpool backward (odd map: last row / column zero)')


@pytest.mark.parametrize('f16', [True, False])
@pytest.mark.parametrize('L,H,W,Cin,Cout,ns,scale', [(2, 6, 5, 64, 64, 1, 1.0), (3, 8, 8, 128, 64, 3, 1.0), (1, 4, 4, 64, 256, 2, 1.0),
                                                     (5, 14, 14, 128, 128, 4, 1e-6), (2, 28, 20, 256, 128, 7, 3e-5),
                                                     (3, 7, 9, 64, 128, 2, 1.0), (2, 31, 17, 128, 128, 5, 1.0),
                                                     (4, 3, 3, 64, 64, 2, 1.0), (1, 70, 66, 64, 64, 9, 1e-4)])
def test_conv3x3_wgrad(hip, f16, L, H, W, Cin, Cout, ns, scale):
    """both arithmetics of the trunk's weight gradient (MMMOT_GEMM_TN: f16x3 = 3-term split on the fp16 matrix cores with a
    device-side power-of-two scale of dZ, f32 = exact fp32 MFMA): gradients of 1e-6 keep fp32-class RELATIVE accuracy"""
    emu = TorchOps(torch.float64)
    dZ, A = rnd(L * H * W, Cout, seed=10) * scale, torch.relu(rnd(L * H * W, Cin, seed=11)) * 2.0
    ref = torch.zeros(ns, 9 * Cout * Cin, dtype=torch.float64)
    emu.conv3x3_wgrad(dZ, A, L, H, W, Cin, Cout, ns, ref)
    got = torch.full((ns, 9 * Cout * Cin), float('nan')).cuda()
    was = hip.tn_f16
    hip.tn_f16 = f16
    try:
        hip.conv3x3_wgrad(dZ.cuda(), A.cuda(), L, H, W, Cin, Cout, ns, got)
    finally:
        hip.tn_f16 = was
    close(got.sum(0) / scale, ref.sum(0).float() / scale, 3e-6, 'conv3x3 weight gradient (%s)' % ('f16x3' if f16 else 'f32'))


def test_conv3x3_first_wgrad(hip):
    emu = TorchOps(torch.float64)
    L, H, W = 3, 9, 6
    dZ, X = rnd(L * H * W, 64, seed=12), rnd(L, 3, H, W, seed=13)
    ref = torch.zeros(1, 64 * 28, dtype=torch.float64)
    emu.conv3x3_first_wgrad(dZ, X, L, H, W, ref)
    got = torch.full((5, 64 * 28), float('nan')).cuda()
    hip.conv3x3_first_wgrad(dZ.cuda(), X.cuda(), L, H, W, got)
    close(got.sum(0), ref[0].float(), 2e-6, 'first-layer weight gradient')


@pytest.mark.parametrize('name', ['s2_C_multiply_none', 's8_S40_C'])
def test_appearance_backward_on_the_device(name):
    """fp32 on the device against float64 autograd: the features to 3e-4; the gradients as far as the non-differentiable
    points of the network allow (tests/test_train_vgg_cpu.py measures the same 1e-3 .. 1e-2 on the fp32 emulation and
    1e-6 on the float64 one: one max-pool argmax or ReLU sign that differs between fp32 and float64) - every tensor within
    5 % in L2, three quarters of them within 0.5 %."""
    c, base = get_case(name)
    m = build_model(c, base, device=DEV)
    dets, info, ds = case_inputs(c)
    plan = m.make_plan([([int(d) for d in ds], None)], c['S'], rows=(0,))
    feats = appearance_autograd(m, plan, dets.to(DEV))
    w = torch.randn(feats.shape, generator=torch.Generator().manual_seed(3))
    (feats * w.to(DEV)).sum().backward()
    mc = build_model(c, base)
    sd = oracle_leaves(mc, ('appearance.',))
    ref = R.appearance(dets.double(), sd, training=True)
    (ref * w.double()).sum().backward()
    assert (feats.detach().cpu().double() - ref.detach()).abs().max().item() < 3e-4
    rel = []
    for k, p in m.named_parameters():
        if not k.startswith('appearance.'):
            continue
        r = sd[k].grad
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if r.norm() < 1e-9:   # convolution biases in front of a BatchNorm: zero gradient
            assert p.grad.norm().item() < 1e-3
            continue
        rel.append(((p.grad.cpu().double() - r).norm() / r.norm()).item())
    rel.sort()
    print('appearance backward on the device %s: relative L2 gradient error median %.1e, worst %.1e over %d tensors' % (
        name, rel[len(rel) // 2], rel[-1], len(rel)))
    assert rel[-1] < 5e-2 and rel[(3 * len(rel)) // 4] < 5e-3


def test_one_full_sgd_step_on_the_device():
    c, base = get_case('s2_B_minus_abs_dual_add')
    m = build_model(c, base, device=DEV)
    dets, info, ds = case_inputs(c)
    counts = [int(d) for d in ds]
    gts = make_gts(counts, 12)
    kw = dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
    lr = 0.02
    mc = build_model(c, base)
    ref_loss, ref_params = sgd_step_reference_full(mc, cfg, kw, dets, info, ds, gts, lr)
    m.train()
    crit = TrackingLoss(**kw)
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    dg = lambda x: [t.to(DEV) for t in x] if isinstance(x, list) else x.to(DEV)
    det, links, new, end, trans = m(dets.to(DEV), {k: v.to(DEV) for k, v in info.items()}, ds)
    loss = crit(ds, dg(gts[0]), dg(gts[1]), dg(gts[2]), dg(gts[3]), det, links, new, end, trans)
    assert abs(loss.item() - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))
    opt.zero_grad()
    loss.backward()
    opt.step()
    errs = []
    for k, p in m.named_parameters():
        if k in ref_params:
            ref = ref_params[k]
            errs.append((p.detach().cpu().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12))
    errs.sort()
    print('one full SGD step on the device: relative parameter difference median %.1e, worst %.1e over %d tensors' % (
        errs[len(errs) // 2], errs[-1], len(errs)))
    assert len(errs) == len(ref_params) and errs[-1] < 5e-3 and errs[len(errs) // 2] < 5e-5
    for bn in (m.appearance.layers[0][1], m.w_det[1]):
        assert int(bn.num_batches_tracked) >= 1


def test_eval_after_training_uses_the_updated_weights():
    """train -> eval: the inference engine's packed weights (folded BatchNorm with the UPDATED running statistics,
    fp16-split copies, folded transforms) are rebuilt after training steps; the eval forward matches the oracle's eval
    forward on the model's current state_dict"""
    c, base = get_case('s2_C_minus_abs_dual_add')
    m = build_model(c, base, device=DEV)
    dets, info, ds = case_inputs(c)
    ddets, dinfo = dets.to(DEV), {k: v.to(DEV) for k, v in info.items()}
    with torch.no_grad():
        before = m(ddets, dinfo, ds)[1][0].clone()   # packs the engine from the initial weights
    counts = [int(d) for d in ds]
    gts = make_gts(counts, 13)
    crit = TrackingLoss(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    dg = lambda x: [t.to(DEV) for t in x] if isinstance(x, list) else x.to(DEV)
    m.train()
    for _ in range(2):
        det, links, new, end, trans = m(ddets, dinfo, ds)
        loss = crit(ds, dg(gts[0]), dg(gts[1]), dg(gts[2]), dg(gts[3]), det, links, new, end, trans)
        opt.zero_grad()
        loss.backward()
        opt.step()
    m.eval()
    with torch.no_grad():
        out = m(ddets, dinfo, ds)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'], neg_threshold=base['neg_threshold'],
               score_arch=base['score_arch'])
    with torch.no_grad():
        ref = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], counts)
    err = max((out[0].cpu() - ref[0]).abs().max().item(), (out[1][0].cpu() - ref[1][0]).abs().max().item(),
              (out[2].cpu() - ref[2]).abs().max().item(), (out[3].cpu() - ref[3]).abs().max().item())
    assert err < 1e-3, err
    assert (out[1][0] - before).abs().max().item() > 1e-4   # and it is not the forward of the initial weights


# ---- the device's training step against the IMPORTED reference's (fixtures of oracle/gen_golden_train.py) ----
from common import compare_train_step, train_case_names  # noqa: E402
from test_train_vgg_cpu import product_train_step  # noqa: E402


@pytest.mark.parametrize('name', train_case_names())
def test_device_training_step_matches_the_reference_fixture(name):
    """scores, loss, gradients (norm of every tensor + element-wise slices) and the BatchNorm buffers of one training step
    on the device against what the imported reference produced for the same sample (tracking_model.py:50-66,
    cost.py:134-185).  Gradient tolerance: the reference's own fp32 noise on the first trunk layers is 0.4 - 0.7 %
    (tests/test_train_vgg_cpu.py); scores within the north-star 1e-3."""
    g, outs, loss, grad_of, buffers = product_train_step(name, device=DEV)
    worst = compare_train_step(g, outs, loss, grad_of, buffers, out_tol=3e-4, loss_tol=1e-4, grad_tol=3e-2, norm_tol=2e-2,
                               bn_tol=5e-5, what=name)
    print('%s: device vs the reference training step: %s' % (name, ' '.join('%s=%.1e' % kv for kv in worst.items())))
