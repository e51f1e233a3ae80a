"""Host logic of the training backward of the pairwise block (mmmot_amd/backward.py) on the torch emulation of the
C-ABI: the launch schedule, the tape, the table building and the gradient algebra must reproduce torch.autograd
through the ORACLE's affinity / new_end / softmax (the CPU restatement of reference modules/gcn.py:68-82,
new_end.py:62-82, tracking_net.py:106-126).  The GPU suite (tests/test_backward_gpu.py) runs the same comparison
through the HIP kernels."""
import pytest
import torch

from common import build_model, get_case
from fake_ops import TorchOps
from mmmot_amd.backward import affinity_autograd, affinity_backward, affinity_forward_train
from oracle import restatement as R


def oracle_block(sd, F3, counts, op, sm):
    """link / new / end of the oracle for a list of frame counts, from F3 [nR, 512, L] (reference layout)"""
    links, news, ends = [], [], []
    start = 0
    for i in range(len(counts) - 1):
        mid, stop = start + counts[i], start + counts[i] + counts[i + 1]
        logit, new, end = R.affinity(F3[:, :, start:mid], F3[:, :, mid:stop], sd, op)
        links.append(R.softmax_mode(logit, sm).squeeze(1))
        news.append(new)
        ends.append(end)
        start = mid
    return links, news, ends


def reference_grads(model, samples, F, op, sm, w_link, w_new, w_end):
    """autograd through the oracle in float64; F [nR, Lt, 512] (ours) <-> [nR, 512, L] per sample (reference)"""
    sd = {k: v.detach().double().clone().requires_grad_(k.startswith('w_link.')) for k, v in model.state_dict().items()}
    Fd = F.detach().double().clone().requires_grad_(True)
    loss = 0.0
    off, lo = 0, 0
    for counts in samples:
        L = sum(counts)
        F3 = Fd[:, off:off + L].permute(0, 2, 1)
        links, news, ends = oracle_block(sd, F3, counts, op, sm)
        d0 = off
        for p, (lk, nw, en) in enumerate(zip(links, news, ends)):
            N, M = counts[p], counts[p + 1]
            n = lk.numel()
            loss = loss + (lk.reshape(-1) * w_link[lo:lo + n].double()).sum()
            lo += n
            loss = loss + (nw * w_new[:, d0 + N:d0 + N + M].double()).sum() + (en * w_end[:, d0:d0 + N].double()).sum()
            d0 += N
        off += L
    loss.backward()
    return Fd.grad, {k: v.grad for k, v in sd.items() if k.startswith('w_link.')}


@pytest.mark.parametrize('op,sm,samples', [
    ('multiply', 'none', [[3, 4]]),
    ('minus_abs', 'dual_add', [[5, 2], [1, 6]]),
    ('minus', 'dual', [[2, 3, 2]]),
    ('multiply', 'dual_max', [[4, 4]]),
    ('minus_abs', 'single', [[1, 1], [3, 5]]),
])
def test_backward_matches_autograd_through_the_oracle(op, sm, samples):
    from mmmot_amd.plan import BatchPlan
    c, base = get_case('s2_C_multiply_none')
    c = dict(c, aff=op, sm=sm)
    if any(len(s) > 2 for s in samples):
        c['counts'] = samples[0]
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')  # exact-fp32 GEMMs in the emulated forward: the comparison is about the gradient algebra
    eng = m.engine()
    plan = BatchPlan([(s, None) for s in samples], 32, 'cpu', use_points=False)
    g = torch.Generator().manual_seed(3)
    F = torch.randn(3, plan.Lt, 512, generator=g) * 0.7
    R_ = plan.pair_tiles.R
    w_link, w_new, w_end = torch.randn(R_, generator=g), torch.randn(3, plan.Lt, generator=g), torch.randn(3, plan.Lt, generator=g)
    link, new, end, tape = affinity_forward_train(eng, plan, F)
    dF, grads = affinity_backward(eng, plan, F, tape, w_link, w_new, w_end)
    dF_ref, g_ref = reference_grads(m, samples, F, op, sm, w_link, w_new, w_end)
    scale = dF_ref.abs().max().item()
    assert (dF.double() - dF_ref).abs().max().item() < 2e-4 * scale, 'dF'
    assert set(grads) == set(g_ref)
    gmax = max(v.abs().max().item() for v in g_ref.values())
    for k, ref in g_ref.items():
        got = grads[k].double().reshape(ref.shape)
        # relative to the parameter's own largest gradient, plus fp32 rounding noise of the whole backward (the bias of
        # a conv that feeds a per-channel GroupNorm has an exactly zero gradient: only noise is left to compare)
        tol = 2e-4 * ref.abs().max().item() + 2e-6 * (1.0 + gmax)
        assert (got - ref).abs().max().item() < tol, (k, (got - ref).abs().max().item(), ref.abs().max().item())


def test_autograd_function_fills_grads():
    from mmmot_amd.plan import BatchPlan
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    plan = BatchPlan([([3, 2], None)], 32, 'cpu', use_points=False)
    F = (torch.randn(3, 5, 512, generator=torch.Generator().manual_seed(1)) * 0.7).requires_grad_(True)
    link, new, end = affinity_autograd(m, plan, F)
    (link.sum() + 2 * new.sum() - end.sum()).backward()
    assert F.grad is not None and F.grad.shape == F.shape and torch.isfinite(F.grad).all()
    for k, p in m.named_parameters():
        if k.startswith('w_link.'):
            assert p.grad is not None and p.grad.shape == p.shape, k
        else:
            assert p.grad is None, k


# ---- second slice: fusion module + training-mode w_det + pairwise block = the whole head -------------------------------
def head_reference(model, counts, cat, fusion, op, sm, w):
    """autograd through the oracle's fusion / affinity and a float64 training-mode w_det (BatchNorm1d on batch statistics,
    no sigmoid: reference tracking_net.py:149-151 with self.training)"""
    import torch.nn.functional as Fn
    heads = ('fusion_module.', 'w_det.', 'w_link.')
    sd = {k: v.detach().double().clone().requires_grad_(k.startswith(heads) and v.dtype.is_floating_point and
                                                        'running' not in k and 'num_batches' not in k)
          for k, v in model.state_dict().items()}
    c = cat.detach().double().clone().requires_grad_(True)
    F3 = R.fusion(c.t().unsqueeze(0), sd, fusion)                    # 3 x 512 x L
    x = F3
    for i, bn in ((0, 1), (3, 4)):
        x = Fn.conv1d(x, sd['w_det.%d.weight' % i], sd['w_det.%d.bias' % i])
        x = Fn.relu(Fn.batch_norm(x, None, None, sd['w_det.%d.weight' % bn], sd['w_det.%d.bias' % bn], True, 0.0, 1e-5))
    det = Fn.conv1d(x, sd['w_det.6.weight'], sd['w_det.6.bias']).squeeze(1)
    links, news, ends = oracle_block(sd, F3, counts, op, sm)
    N, M = counts
    loss = (det * w['det'].double()).sum() + (links[0].reshape(-1) * w['link'].double()).sum() + \
        (news[0] * w['new'][:, N:].double()).sum() + (ends[0] * w['end'][:, :N].double()).sum()
    loss.backward()
    return c.grad, {k: v.grad for k, v in sd.items() if v.requires_grad}, det.detach()


@pytest.mark.parametrize('fusion,op,sm', [('A', 'multiply', 'none'), ('B', 'minus_abs', 'dual_add'), ('C', 'multiply', 'none'),
                                          ('C', 'minus_abs', 'dual_add')])
def test_head_backward_matches_autograd(fusion, op, sm):
    from mmmot_amd.backward import head_backward, head_forward_train
    from mmmot_amd.plan import BatchPlan
    c, base = get_case('s2_C_multiply_none')
    c = dict(c, fusion=fusion, aff=op, sm=sm)
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    eng = m.engine()
    counts = [4, 5]
    plan = BatchPlan([(counts, None)], 32, 'cpu', use_points=False)
    g = torch.Generator().manual_seed(5)
    cat = torch.randn(plan.Lt, 1024, generator=g) * 0.8
    w = dict(det=torch.randn(3, plan.Lt, generator=g), link=torch.randn(plan.pair_tiles.R, generator=g),
             new=torch.randn(3, plan.Lt, generator=g), end=torch.randn(3, plan.Lt, generator=g))
    det, link, new, end, tape = head_forward_train(eng, m, plan, cat)
    dcat, grads = head_backward(eng, m, plan, cat, tape, w['det'], w['link'], w['new'], w['end'])
    dcat_ref, g_ref, det_ref = head_reference(m, counts, cat, fusion, op, sm, w)
    assert (det.double() - det_ref).abs().max().item() < 2e-4  # training-mode scores: raw, batch-statistics BatchNorm
    assert (dcat.double() - dcat_ref).abs().max().item() < 3e-4 * dcat_ref.abs().max().item(), 'dcat'
    assert set(grads) == set(g_ref), sorted(set(grads) ^ set(g_ref))
    gmax = max(v.abs().max().item() for v in g_ref.values())
    for k, ref in g_ref.items():
        got = grads[k].double().reshape(ref.shape)
        tol = 3e-4 * ref.abs().max().item() + 3e-6 * (1.0 + gmax)
        assert (got - ref).abs().max().item() < tol, (k, (got - ref).abs().max().item(), ref.abs().max().item())


def test_head_autograd_updates_batchnorm_buffers_like_torch():
    from mmmot_amd.backward import head_autograd
    from mmmot_amd.plan import BatchPlan
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f32')
    plan = BatchPlan([([3, 4], None)], 32, 'cpu', use_points=False)
    cat = (torch.randn(plan.Lt, 1024, generator=torch.Generator().manual_seed(2)) * 0.8).requires_grad_(True)
    rm0, rv0 = m.w_det[1].running_mean.clone(), m.w_det[1].running_var.clone()
    det, link, new, end = head_autograd(m, plan, cat)
    (det.sum() + link.sum() + new.sum() - end.sum()).backward()
    assert cat.grad is not None and torch.isfinite(cat.grad).all()
    for k, p in m.named_parameters():
        assert (p.grad is not None) == (k.split('.')[0] in ('fusion_module', 'w_det', 'w_link')), k
    # running statistics: the momentum update of torch's training-mode BatchNorm1d on the same conv output
    import torch.nn.functional as Fn
    F3 = R.fusion(cat.detach().t().unsqueeze(0), {k: v.detach() for k, v in m.state_dict().items()}, 'C')
    x = Fn.conv1d(F3, m.w_det[0].weight.detach(), m.w_det[0].bias.detach())
    rm, rv = rm0.clone(), rv0.clone()
    Fn.batch_norm(x, rm, rv, None, None, True, 0.1, 1e-5)
    assert torch.allclose(m.w_det[1].running_mean, rm, atol=1e-5) and torch.allclose(m.w_det[1].running_var, rv, atol=1e-5)


def test_refresh_head_repacks_only_the_head():
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    eng = m.engine()
    vgg_before, pn_before = eng.P['vgg'], eng.P['pointnet']
    wa_before = eng.P['w_link']['wa'].clone()
    with torch.no_grad():
        m.w_link.conv1[0].weight.mul_(1.5)
    assert m.refresh_head() is eng and eng.P['vgg'] is vgg_before and eng.P['pointnet'] is pn_before
    wa = eng.P['w_link']['wa']
    assert torch.allclose(wa[512:], wa_before[512:] * 1.5) and torch.equal(wa[:512], wa_before[:512])
    assert 'wa_h16' in eng.P['w_link']  # the fp16-split copies are rebuilt too


def test_two_level_segment_sums_equal_the_direct_sums():
    """long strided sums (the per-tile partials of a full-resolution trunk layer: thousands of rows per segment) are
    summed in two levels - chunks, then chunk sums; short ones keep their single table"""
    import numpy as np
    from mmmot_amd.backward import SEGSUM_CHUNK, _chunked_segments, _colsum, _segsum
    c, base = get_case('s2_C_multiply_none')
    eng = build_model(c, base, ops=TorchOps(torch.float64)).engine()
    g = torch.Generator().manual_seed(5)
    T, C = 5 * SEGSUM_CHUNK + 37, 8
    X = torch.randn(2 * T + 6, C, generator=g, dtype=torch.float64).float()
    segs = _chunked_segments(np.array([0, 1, 2 * T]), np.array([T, T, 3]), 2, 'cpu')
    assert isinstance(segs, tuple) and segs[0].n == 2 * 6 + 1 and segs[1].n == 3
    out = torch.zeros(3, C)
    _segsum(eng, X, C, segs, out)
    want = torch.stack([X[0:2 * T:2].double().sum(0), X[1:2 * T:2].double().sum(0), X[2 * T:2 * T + 6:2].double().sum(0)])
    assert (out.double() - want).abs().max().item() < 1e-4
    short = _chunked_segments(np.array([0, 1]), np.array([40, 40]), 2, 'cpu')
    assert not isinstance(short, tuple)
    tall = torch.randn(9000, C, generator=g)
    assert (_colsum(eng, tall).double() - tall.double().sum(0)).abs().max().item() < 1e-3
    # 9000 x 8 is viewed as 1125 rows of 64 (three halvings, 1125 is odd), summed as 9 chunks of 128 rows, then the 9 chunk
    # sums, then the eight row classes are folded
    assert eng._colsum_segs[(1125, 'cpu')].n == 9 and eng._colsum_segs[(9, 'cpu')].n == 1
    assert (9000, 'cpu') not in eng._colsum_segs
    odd = torch.randn(1001, 12, generator=g)                      # odd row count: no folding, 8 chunks, then one segment
    assert (_colsum(eng, odd).double() - odd.double().sum(0)).abs().max().item() < 1e-3
    assert eng._colsum_segs[(1001, 'cpu')].n == 8
    wide = torch.randn(4096, 520, generator=g)[:, :512]           # a column slice (not contiguous): summed as it lies
    assert (_colsum(eng, wide).double() - wide.double().sum(0)).abs().max().item() < 2e-3
    small = torch.randn(7, 16, generator=g)
    assert (_colsum(eng, small).double() - small.double().sum(0)).abs().max().item() < 1e-5
