"""Training-mode image encoder (mmmot_amd/train_vgg.py) on the torch emulation of the C-ABI: features and every
``appearance.*`` gradient against torch.autograd through the oracle's training-mode trunk (float64), the BatchNorm2d
buffer updates against torch's, and one SGD step on the WHOLE network against the oracle's.  tests/test_train_vgg_gpu.py
runs the same through the HIP kernels."""
import pytest
import torch

from common import build_model, case_inputs, get_case
from fake_ops import TorchOps
from mmmot_amd import TrackingLoss
from mmmot_amd.train_vgg import appearance_autograd
from oracle import restatement as R
from test_train_cpu import make_gts


def oracle_leaves(model, prefixes, dtype=torch.float64):
    train = {k for k, p in model.named_parameters() if p.requires_grad and k.startswith(prefixes)}
    return {k: (v.detach().to(dtype).clone().requires_grad_(k in train) if v.dtype.is_floating_point else v.detach().clone())
            for k, v in model.state_dict().items()}


def check_grads(model, sd, prefix, rtol=3e-4):
    gmax = max(v.grad.abs().max().item() for k, v in sd.items() if k.startswith(prefix) and v.grad is not None)
    seen, worst = 0, 0.0
    for k, p in model.named_parameters():
        if not k.startswith(prefix):
            continue
        ref = sd[k].grad
        assert ref is not None and p.grad is not None, k
        seen += 1
        err = (p.grad.detach().cpu().double() - ref).abs().max().item()
        assert err < rtol * ref.abs().max().item() + 3e-6 * (1.0 + gmax), (k, err, ref.abs().max().item())
        worst = max(worst, err / (ref.abs().max().item() + 1e-12))
    return seen, worst


# The emulation runs every operator in float64 here (tensors between operators stay fp32): the schedule, the tape and the
# gradient algebra must then agree with float64 autograd to ~1e-6.  In fp32 (the device) the same comparison is limited by
# the non-differentiable points of the network, not by arithmetic: with 2x2 .. 4x4 feature maps and a dozen crops a
# max-pool argmax or a ReLU sign that differs between fp32 and float64 in ONE window moves a gradient tensor by 1e-2
# (measured: conv1_1 .. conv4_2 1e-3 .. 1e-2, conv4_3 .. conv5_3 1e-5 in the same run) - test_train_vgg_gpu.py bounds that.
@pytest.mark.parametrize('name', ['s2_C_multiply_none', 's8_S40_C'])
def test_appearance_backward_matches_autograd_through_the_oracle(name):
    c, base = get_case(name)
    m = build_model(c, base, ops=TorchOps(torch.float64))
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    plan = m.make_plan([([int(d) for d in ds], None)], c['S'], rows=(0,))
    rm0 = m.appearance.layers[1][1].running_mean.clone()
    bn0 = m.appearance.layers[0][1]
    rm, rv = bn0.running_mean.clone(), bn0.running_var.clone()   # the buffers before the training-mode forward
    nb0 = int(bn0.num_batches_tracked)
    feats = appearance_autograd(m, plan, dets)
    w = torch.randn(feats.shape, generator=torch.Generator().manual_seed(3))
    (feats * w).sum().backward()
    sd = oracle_leaves(m, ('appearance.',))
    ref = R.appearance(dets.double(), sd, training=True)
    (ref * w.double()).sum().backward()
    assert (feats.detach().double() - ref.detach()).abs().max().item() < 2e-5
    seen, worst = check_grads(m, sd, 'appearance.', rtol=2e-5)
    assert seen == 13 * 4 + 4 * 10
    print('appearance backward %s: worst relative gradient error %.1e over %d tensors' % (name, worst, seen))
    # running statistics: torch's training-mode BatchNorm2d momentum update on the same pre-BatchNorm tensor
    import torch.nn.functional as Fn
    x = Fn.conv2d(dets, m.appearance.layers[0][0].weight.detach(), m.appearance.layers[0][0].bias.detach(), padding=1)
    Fn.batch_norm(x, rm, rv, None, None, True, 0.1, 1e-5)
    bn = m.appearance.layers[0][1]
    assert torch.allclose(bn.running_mean, rm, atol=1e-5) and torch.allclose(bn.running_var, rv, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == nb0 + 1 and not torch.equal(m.appearance.layers[1][1].running_mean, rm0)


def sgd_step_reference_full(model, cfg, kw, dets, info, ds, gts, lr):
    """one SGD step on EVERY parameter through the oracle in float64 (training-mode trunk included)"""
    sd = oracle_leaves(model, ('appearance.', 'point_net.', 'fusion_module.', 'w_det.', 'w_link.'))
    det, links, new, end, trans = R.tracking_forward_train(sd, cfg, None, info['points'].double(), info['points_split'], ds,
                                                           crops=dets.double())
    gt_det, gt_link, gt_new, gt_end = gts
    loss = R.tracking_loss([int(d) for d in ds], gt_det.double(), [g.double() for g in gt_link], gt_new.double(),
                           gt_end.double(), det, links, new, end, trans, **kw)
    loss.backward()
    return loss.item(), {k: (v.detach() - lr * v.grad) for k, v in sd.items() if v.dtype.is_floating_point and v.grad is not None}


def test_one_full_sgd_step_matches_the_oracle():
    """tracking_model.py:50-66 end to end: training-mode forward of the whole network (batch-statistics BatchNorm in the
    trunk and w_det) -> TrackingLoss -> backward -> SGD step; every updated parameter against the oracle's"""
    c, base = get_case('s2_B_minus_abs_dual_add')
    m = build_model(c, base, ops=TorchOps(torch.float64))
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    counts = [int(d) for d in ds]
    gts = make_gts(counts, 12)
    kw = dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
    lr = 0.02
    ref_loss, ref_params = sgd_step_reference_full(m, cfg, kw, dets, info, ds, gts, lr)
    m.train()
    crit = TrackingLoss(**kw)
    crit.ops = TorchOps(torch.float64)
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    det, links, new, end, trans = m(dets, info, ds)
    loss = crit(ds, gts[0], gts[1], gts[2], gts[3], det, links, new, end, trans)
    assert abs(loss.item() - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    opt.zero_grad()
    loss.backward()
    opt.step()
    worst, bad = 0.0, []
    for k, p in m.named_parameters():
        if k in ref_params:
            ref = ref_params[k]
            err = (p.detach().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
            worst = max(worst, err)
            if err >= 2e-5:
                bad.append((k, '%.1e' % err))
    assert not bad, bad
    assert len([k for k in ref_params if k.startswith('appearance.')]) == 92
    print('one full SGD step: worst relative parameter difference %.2e over %d tensors' % (worst, len(ref_params)))


# ---- the product's training step against the IMPORTED reference's (fixtures of oracle/gen_golden_train.py) ----
from common import compare_train_step, load_train_case, manifest, train_case_names  # noqa: E402


def product_train_step(name, device='cpu', ops=None):
    """model.train() forward -> TrackingLoss -> backward of the product on the fixture's sample"""
    c, kw, g, (dets, info, ds), gts = load_train_case(name)
    m = build_model(c, manifest()['base_kwargs'], device=device, ops=ops)
    m.train()
    crit = TrackingLoss(**kw)
    if ops is not None:
        m.set_trunk('f32')
        crit.ops = ops
    to = lambda x: [t.to(device) for t in x] if isinstance(x, list) else x.to(device)
    if 'rng_seed' in c:  # DropBlock fixtures: the seed masks come from the global host generator, like the reference's
        torch.manual_seed(c['rng_seed'])
    det, links, new, end, trans = m(to(dets), {k: to(v) for k, v in info.items()}, ds)
    loss = crit(ds, to(gts[0]), to(gts[1]), to(gts[2]), to(gts[3]), det, links, new, end, trans)
    loss.backward()
    params = dict(m.named_parameters())
    buffers = dict(m.named_buffers())
    return g, (det, links, new, end, trans), loss.item(), (lambda k: params[k].grad), buffers


@pytest.mark.parametrize('name', train_case_names())
def test_training_step_matches_the_reference_fixture(name):
    """host logic + float64 emulation of the C-ABI: scores, loss, gradients and the BatchNorm buffers of one training step
    against what the imported reference produced (tracking_model.py:50-66, cost.py:134-185)"""
    g, outs, loss, grad_of, buffers = product_train_step(name, ops=TorchOps(torch.float64))
    # gradients: the reference's fp32 backward through 13 batch-normalised layers carries 0.4 - 0.7 % of rounding noise
    # on the first layers' gradients (this float64 run and the reference differ by that much on appearance.layers.0.*;
    # the fp32 oracle, which repeats the reference's operations, agrees with it to 2e-5: tests/test_train_oracle.py)
    worst = compare_train_step(g, outs, loss, grad_of, buffers, out_tol=5e-5, loss_tol=5e-6, grad_tol=2e-2, norm_tol=1e-2,
                               bn_tol=1e-5, what=name)
    print('%s: product (emulated C-ABI) vs the reference training step: %s' % (name, ' '.join('%s=%.1e' % kv for kv in worst.items())))


def test_weight_gradient_shares_fill_the_chip_once():
    """_wgrad_shares: the f16x3 weight-gradient kernel runs one workgroup per (channel-tile pair, tap row, share) and one
    (two with 64 x 64 tiles) per CU - the share count gives at most one round of workgroups on 256 CUs, at least one share,
    at most the 256 the C entry point accepts, and never more shares than 256-pixel runs of the layer"""
    from mmmot_amd.train_vgg import _wgrad_shares
    dev = torch.device('cpu')  # no device properties on the CPU: 256 CUs assumed
    for cin, cout, rows in [(64, 64, 1103872), (64, 128, 275968), (128, 128, 275968), (128, 256, 68992),
                            (256, 256, 68992), (256, 512, 17248), (512, 512, 17248), (512, 512, 4312), (512, 512, 200)]:
        ns = _wgrad_shares(dev, cin, cout, rows, True)
        tn, tk = (128 if cout % 128 == 0 else 64), (128 if cin % 128 == 0 else 64)
        wgs = (cout // tn) * (cin // tk) * 3 * ns
        slots = 256 * (2 if (tn, tk) == (64, 64) else 1)
        assert 1 <= ns <= 256 and ns <= max(1, rows // 256)
        assert wgs <= slots or ns == 1, (cin, cout, rows, ns)
        if rows // 256 >= slots:                       # enough pixels: the single round is nearly full
            assert wgs > slots - (cout // tn) * (cin // tk) * 3
    assert _wgrad_shares(dev, 512, 512, 17248, False) == 2  # fp32 kernel: the round-3 rule (64 x 64 tiles per tap)
