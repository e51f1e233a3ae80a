"""Host logic of the training step's second slice (mmmot_amd/train.py) on the torch emulation of the C-ABI: the
TrackingLoss operator against the fixtures of the IMPORTED reference, the PointNet backward and one whole SGD step
against torch.autograd through the oracle in float64.  tests/test_train_gpu.py runs the same through the HIP kernels."""
import pytest
import torch

from common import build_model, case_inputs, get_case
from fake_ops import TorchOps
from mmmot_amd import TrackingLoss, build_criterion
from mmmot_amd.train import FOLDED, fold_pointnet, pointnet_autograd
from oracle import restatement as R
from test_train_oracle import CASES, load_case, load_ghm_sequence

HEADS = ('point_net.', 'fusion_module.', 'w_det.', 'w_link.')


@pytest.mark.parametrize('name', CASES)
def test_tracking_loss_operator_matches_the_reference_fixture(name):
    counts, kw, ins, ref = load_case(name)
    crit = TrackingLoss(**kw)
    crit.ops = TorchOps(torch.float64)
    leaf = lambda x: x.clone().requires_grad_(True)
    det, new, end = leaf(ins['det']), leaf(ins['new']), leaf(ins['end'])
    links, trans = [leaf(l) for l in ins['links']], [leaf(x) for x in ins['trans']]
    loss = crit([torch.tensor(c) for c in counts], ins['gt_det'], ins['gt_link'], ins['gt_new'], ins['gt_end'], det, links,
                new, end, trans)
    loss.backward()
    assert abs(loss.item() - ref['loss']) < 2e-6
    for got, want in [(det, ref['det']), (new, ref['new']), (end, ref['end'])] + list(zip(links, ref['links'])) + \
            list(zip(trans, ref['trans'])):
        g = got.grad if got.grad is not None else torch.zeros_like(got)
        assert (g - want).abs().max().item() < 1e-6


def test_criterion_constructor_mirrors_the_reference():
    with pytest.raises(AssertionError):      # cost.py:73: the reference's own default link loss type fails its assert
        TrackingLoss()
    with pytest.raises(NotImplementedError):
        TrackingLoss(detloss_type='focal', linkloss_type='l2')
    c = build_criterion(dict(det_loss='bce', link_loss='l2', smooth_ratio=0, det_ratio=1.5, trans_ratio=0.001, trans_last=False))
    assert c.det_ratio == 1.5 and c.trans_ratio == 0.001 and c.linkloss_type == 'l2'


def run_ghm_sequence(crit, dev, loss_tol, grad_tol):
    """the 'ghm' loss types keep running per-bin counts: one criterion, the fixture's three consecutive samples"""
    kw, steps = load_ghm_sequence()
    for st in steps:
        ins = st['ins']
        leaf = lambda x: x.clone().to(dev).requires_grad_(True)
        d = lambda x: x.to(dev)
        det, new, end, link = leaf(ins['det']), leaf(ins['new']), leaf(ins['end']), leaf(ins['links'][0])
        loss = crit([torch.tensor(c) for c in st['counts']], d(ins['gt_det']), [d(ins['gt_link'][0])], d(ins['gt_new']),
                    d(ins['gt_end']), det, [link], new, end, [d(x) for x in ins['trans']])
        loss.backward()
        assert abs(loss.item() - st['loss']) < loss_tol * max(1.0, abs(st['loss']))
        for got, key in ((det, 'det'), (new, 'new'), (end, 'end'), (link, 'link')):
            assert (got.grad.cpu() - st['grads'][key]).abs().max().item() < grad_tol
        for which in ('det', 'end'):  # the module's state after the step = the reference's acc_sum lists
            assert (crit.ghm_state(which).cpu() - st['acc'][which]).abs().max().item() < 1e-9


def test_ghm_loss_operator_sequence_matches_the_reference_fixture():
    kw, _ = load_ghm_sequence()
    crit = TrackingLoss(**kw)
    crit.ops = TorchOps(torch.float64)
    run_ghm_sequence(crit, 'cpu', 2e-6, 1e-6)


def oracle_sd(model, dtype=torch.float64):
    """reference-keyed float64 leaves; trainable exactly where the model's parameter is (``idt`` is a frozen Parameter)"""
    train = {k for k, p in model.named_parameters() if p.requires_grad and k.startswith(HEADS)}
    return {k: (v.detach().to(dtype).clone().requires_grad_(k in train)
                if v.dtype.is_floating_point else v.detach().clone()) for k, v in model.state_dict().items()}


def test_pointnet_backward_matches_autograd_through_the_oracle():
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    ps = info['points_split'].reshape(-1).long()
    points = info['points'].reshape(-1, 3).contiguous()
    plan = m.make_plan([([int(d) for d in ds], ps.numpy())], c['S'])
    out, trans = pointnet_autograd(m, plan, points)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    wt = torch.randn(64, 64, generator=torch.Generator().manual_seed(4))
    ((out * w).sum() + (trans[1][0] * wt).sum()).backward()
    sd = oracle_sd(m)
    ref_out, ref_trans = R.pointnet(info['points'].double().transpose(-1, -2), ps, sd)
    ((ref_out * w.double()).sum() + (ref_trans[1][0] * wt.double()).sum()).backward()
    assert (out.detach().double() - ref_out.detach()).abs().max().item() < 2e-4
    gmax = max(v.grad.abs().max().item() for k, v in sd.items() if k.startswith('point_net.') and v.grad is not None)
    seen = 0
    for k, p in m.named_parameters():
        if not k.startswith('point_net.'):
            assert p.grad is None, k
            continue
        ref = sd[k].grad
        if ref is None:   # STN trunk, avg_bn: never reach the output (zero gradient in the reference, none here)
            assert p.grad is None, k
            continue
        seen += 1
        err = (p.grad.double() - ref).abs().max().item()
        assert err < 3e-4 * ref.abs().max().item() + 3e-6 * (1.0 + gmax), (k, err, ref.abs().max().item())
    assert seen >= len(FOLDED) - 1


def test_fold_matches_the_pack_time_fold():
    from mmmot_amd.pack import pack_weights
    c, base = get_case('s7_refl_B')   # 4-channel input: 4 x 4 first transform
    m = build_model(c, base, ops=TorchOps())
    W, trans = fold_pointnet(m.point_net)
    P = pack_weights(m.state_dict(), 'B', 'cpu')['pointnet']
    for k in ('w1', 'w2', 'wc1a', 'wc1b', 'w5'):
        assert torch.allclose(W[k].detach(), P[k], atol=2e-6), k
    assert torch.allclose(trans[0][0].detach(), P['trans1'], atol=1e-6) and trans[0].shape == (1, 4, 4)


def sgd_step_reference(model, cfg, kw, dets, info, ds, gts, lr, img=None):
    """one SGD step on PointNet + head through the ORACLE in float64.  Image features (frozen, an INPUT of the trained
    part): the oracle's eval-mode trunk, or ``img`` - the features the product's own trunk produced, so that the
    comparison isolates the trained part (the trunk has its own parity tests; with the 9-detection GroupNorms of the
    small fixtures a 1e-6 difference of the features is amplified a hundredfold in some channels' gradients)."""
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        img = R.appearance(dets, sd32).double() if img is None else img.detach().cpu().double()
    sd = oracle_sd(model)
    det, links, new, end, trans = R.tracking_forward_train(sd, cfg, img, info['points'].double(), info['points_split'], ds)
    gt_det, gt_link, gt_new, gt_end = gts
    loss = R.tracking_loss([int(d) for d in ds], gt_det.double(), [g.double() for g in gt_link], gt_new.double(),
                           gt_end.double(), det, links, new, end, trans, **kw)
    loss.backward()
    new_params = {k: (v.detach() - lr * v.grad) for k, v in sd.items() if v.dtype.is_floating_point and v.grad is not None}
    return loss.item(), new_params, (det.detach(), [l.detach() for l in links], new.detach(), end.detach())


def make_gts(counts, seed):
    g = torch.Generator().manual_seed(seed)
    L = sum(counts)
    gt_det = (torch.rand(L, generator=g) > 0.3).float()
    gt_new, gt_end = (torch.rand(L, generator=g) > 0.6).float(), (torch.rand(L, generator=g) > 0.6).float()
    gt_link = [(torch.rand(1, counts[i], counts[i + 1], generator=g) > 0.8).float() for i in range(len(counts) - 1)]
    return gt_det, gt_link, gt_new, gt_end


@pytest.mark.parametrize('name', ['s2_C_multiply_none', 's2_A_minus_abs_dual_add'])
def test_one_sgd_step_matches_the_oracle(name):
    c, base = get_case(name)
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    counts = [int(d) for d in ds]
    gts = make_gts(counts, 11)
    kw = dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
    lr = 0.05
    ref_loss, ref_params, ref_scores = sgd_step_reference(m, cfg, kw, dets, info, ds, gts, lr)
    m.train()
    m.freeze_appearance = True               # this test: PointNet + head on frozen eval-mode image features
    crit = TrackingLoss(**kw)
    crit.ops = TorchOps()
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    det, links, new, end, trans = m(dets, info, ds)
    assert det.shape == (3, sum(counts)) and new.shape == (3, sum(counts) - counts[0]) and end.shape == (3, sum(counts) - counts[-1])
    assert (det.detach().double() - ref_scores[0]).abs().max().item() < 3e-4
    assert (links[0].detach().double() - ref_scores[1][0]).abs().max().item() < 3e-4
    loss = crit(ds, *gts[:1], gts[1], gts[2], gts[3], det, links, new, end, trans)
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    opt.zero_grad()
    loss.backward()
    opt.step()
    worst = 0.0
    for k, p in m.named_parameters():
        if k in ref_params:
            ref = ref_params[k]
            err = (p.detach().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
            worst = max(worst, err)
            assert err < 1e-5, (k, err)
        else:
            assert p.grad is None or not k.startswith(HEADS), k
    assert len(ref_params) >= 76  # fusion A has fewer tensors than C
    print('one SGD step, worst relative parameter difference: %.2e over %d tensors' % (worst, len(ref_params)))
    # the engine notices the optimizer step: the next forward runs on the updated head
    assert not m.head_is_current()
    m(dets, info, ds)
    packed = m.engine().P['w_link']['w3']
    assert torch.equal(packed, m.w_link.conv1[3].weight.detach().flatten(1))  # re-packed from the stepped parameters


def test_three_sgd_steps_match_the_oracle():
    """Several steps in a row: the folded PointNet weights are fresh tensors every step, freed after the
    backward, and the allocator hands their addresses to the next step's fold - a transposed-weight cache keyed by
    address alone then serves the PREVIOUS step's weights to the input-gradient GEMM from step 2 on (1e-2 off).
    Every step is compared with the oracle's step FROM THE MODEL'S OWN STATE: on this 9-detection fixture the
    normalisations amplify a 6e-7 difference of the parameters into 20 % of some gradients after one step of this size, so
    two trajectories cannot be compared - one step at a time can."""
    name = 's2_C_multiply_none'
    c, base = get_case(name)
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    counts = [int(d) for d in ds]
    kw = dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
    lr, steps = 0.05, 3
    m.train()
    m.freeze_appearance = True
    crit = TrackingLoss(**kw)
    crit.ops = TorchOps()
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    worst = 0.0
    for it in range(steps):
        gts = make_gts(counts, 20 + it)
        _, ref_params, _ = sgd_step_reference(m, cfg, kw, dets, info, ds, gts, lr)
        det, links, new, end, trans = m(dets, info, ds)
        loss = crit(ds, gts[0], gts[1], gts[2], gts[3], det, links, new, end, trans)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, p in m.named_parameters():
            if k in ref_params:
                ref = ref_params[k]
                err = (p.detach().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
                worst = max(worst, err)
                assert err < 3e-5, (it, k, err)
    print('three SGD steps, worst relative parameter difference of a step: %.2e' % worst)


@pytest.mark.parametrize('what', ['one_step', 'encoder_only'])
def test_eval_after_training_repacks(what):
    """eval -> training step(s) -> eval must compute with the UPDATED weights: after exactly one step (the
    flag refresh_head_device sets is only reached by the NEXT training forward), and after training that leaves the head
    untouched (PointNet-only: the head's parameter versions never change)."""
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    dets, info, ds = case_inputs(c)
    counts = [int(d) for d in ds]
    m.eval()
    with torch.no_grad():
        before = m(dets, info, ds)
    m.train()
    m.freeze_appearance = True
    crit = TrackingLoss(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    crit.ops = TorchOps()
    params = [p for k, p in m.named_parameters() if (what == 'one_step' or k.startswith('point_net.'))]
    opt = torch.optim.SGD(params, lr=0.05)
    gts = make_gts(counts, 31)
    det, links, new, end, trans = m(dets, info, ds)
    loss = crit(ds, gts[0], gts[1], gts[2], gts[3], det, links, new, end, trans)
    opt.zero_grad()
    loss.backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        after = m(dets, info, ds)
        m.invalidate()
        fresh = m(dets, info, ds)
    assert (after[1][0] - before[1][0]).abs().max().item() > 1e-4, 'the eval forward still runs on the old weights'
    for a, f in zip((after[0], after[1][0], after[2], after[3]), (fresh[0], fresh[1][0], fresh[2], fresh[3])):
        assert torch.equal(a, f)


def test_transposed_weight_cache_survives_address_reuse():
    """dgrad_gemm caches W^T per (address, version, shape).  A freed weight's address is handed to the next tensor of the
    same size (version 0 again): the cache must not serve the old transpose for it.  An entry therefore pins
    its weight - the address cannot come back while the entry lives."""
    from mmmot_amd.backward import dgrad_gemm
    from mmmot_amd.plan import RowTiles
    c, base = get_case('s2_C_multiply_none')
    eng = build_model(c, base, ops=TorchOps()).engine()
    tiles = RowTiles([8], 'cpu')
    g = torch.Generator().manual_seed(3)
    X = torch.randn(8, 16, generator=g)
    for attempt in range(40):
        W1 = torch.randn(16, 24, generator=g)
        ptr = W1.data_ptr()
        Y1 = torch.empty(8, 24)
        dgrad_gemm(eng, W1, tiles, X, Y1)
        assert torch.allclose(Y1, X @ W1, atol=1e-5)
        del W1
        W2 = torch.randn(16, 24, generator=g)   # without the pin: usually the address W1 just gave back
        Y2 = torch.empty(8, 24)
        dgrad_gemm(eng, W2, tiles, X, Y2)
        assert torch.allclose(Y2, X @ W2, atol=1e-5), 'stale transposed weight served for a new tensor at a reused address'
        assert W2.data_ptr() != ptr
    assert len(eng._wt_cache) <= 64
    # least-recently-USED eviction: a weight that is hit on every step survives any number of one-shot
    # weights passing through; the round-4 first-in-first-out cache dropped it after 32 insertions
    keep = torch.randn(16, 24, generator=g)
    dgrad_gemm(eng, keep, tiles, X, torch.empty(8, 24))
    kept_t = eng._wt_cache[(keep.data_ptr(), keep._version, 16, 24)][1]
    for _ in range(200):
        dgrad_gemm(eng, torch.randn(16, 24, generator=g), tiles, X, torch.empty(8, 24))
        dgrad_gemm(eng, keep, tiles, X, torch.empty(8, 24))
    assert eng._wt_cache[(keep.data_ptr(), keep._version, 16, 24)][1] is kept_t and len(eng._wt_cache) <= 64
