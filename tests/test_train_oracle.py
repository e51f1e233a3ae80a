"""The oracle's TrackingLoss restatement (oracle/restatement.py::tracking_loss, reference cost.py:134-185) against the
fixtures oracle/gen_golden_loss.py made by running the IMPORTED reference: loss value and every gradient."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from common import GOLD
from oracle import restatement as R

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, 'loss_*.npz')))


def load_case(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    counts = [int(c) for c in g['counts']]
    kw = dict(ast.literal_eval(str(g['kwargs'])))
    t = lambda k: torch.from_numpy(g[k])
    n = len(counts) - 1
    ins = dict(det=t('det'), new=t('new'), end=t('end'), links=[t('link%d' % i) for i in range(n)],
               trans=[t('trans0'), t('trans1')], gt_det=t('gt_det'), gt_new=t('gt_new'), gt_end=t('gt_end'),
               gt_link=[t('gt_link%d' % i) for i in range(n)])
    ref = dict(loss=float(g['loss']), det=t('g_det'), new=t('g_new'), end=t('g_end'),
               links=[t('g_link%d' % i) for i in range(n)], trans=[t('g_trans0'), t('g_trans1')])
    return counts, kw, ins, ref


def test_fixtures_exist():
    assert len(CASES) >= 4


@pytest.mark.parametrize('name', CASES)
def test_oracle_loss_matches_the_reference_fixture(name):
    counts, kw, ins, ref = load_case(name)
    leaf = lambda x: x.clone().requires_grad_(True)
    det, new, end = leaf(ins['det']), leaf(ins['new']), leaf(ins['end'])
    links, trans = [leaf(l) for l in ins['links']], [leaf(x) for x in ins['trans']]
    loss = R.tracking_loss(counts, ins['gt_det'], ins['gt_link'], ins['gt_new'], ins['gt_end'], det, links, new, end, trans, **kw)
    loss.backward()
    assert abs(loss.item() - ref['loss']) < 1e-6
    for got, want in [(det, ref['det']), (new, ref['new']), (end, ref['end'])] + list(zip(links, ref['links'])) + \
            list(zip(trans, ref['trans'])):
        g = got.grad if got.grad is not None else torch.zeros_like(got)
        assert (g - want).abs().max().item() < 1e-7


def load_ghm_sequence():
    """tests/golden/lossseq_ghm.npz (oracle/gen_golden_loss.py::ghm_sequence): ONE criterion of the imported reference
    with detloss_type = endloss_type = 'ghm' called on three consecutive samples - the loss keeps running per-bin counts"""
    g = np.load(os.path.join(GOLD, 'lossseq_ghm.npz'))
    kw = dict(ast.literal_eval(str(g['kwargs'])))
    steps = []
    for k in range(int(g['steps'])):
        t = lambda key: torch.from_numpy(g['%s_%d' % (key, k)])
        steps.append(dict(counts=[int(c) for c in g['counts_%d' % k]], loss=float(g['loss_%d' % k]),
                          ins=dict(det=t('det'), new=t('new'), end=t('end'), links=[t('link0')], trans=[t('trans0'), t('trans1')],
                                   gt_det=t('gt_det'), gt_new=t('gt_new'), gt_end=t('gt_end'), gt_link=[t('gt_link0')]),
                          grads=dict(det=t('g_det'), new=t('g_new'), end=t('g_end'), link=t('g_link0')),
                          acc=dict(det=t('acc_det'), end=t('acc_end'))))
    return kw, steps


def test_oracle_ghm_loss_sequence_matches_the_reference_fixture():
    kw, steps = load_ghm_sequence()
    state = dict(det=R.new_ghm_state(), end=R.new_ghm_state())
    for st in steps:
        ins = st['ins']
        leaf = lambda x: x.clone().requires_grad_(True)
        det, new, end, link = leaf(ins['det']), leaf(ins['new']), leaf(ins['end']), leaf(ins['links'][0])
        loss = R.tracking_loss(st['counts'], ins['gt_det'], ins['gt_link'], ins['gt_new'], ins['gt_end'], det, [link], new, end,
                               [x.clone() for x in ins['trans']], ghm_state=state, **kw)
        loss.backward()
        assert abs(loss.item() - st['loss']) < 1e-6
        for got, key in ((det, 'det'), (new, 'new'), (end, 'end'), (link, 'link')):
            assert (got.grad - st['grads'][key]).abs().max().item() < 1e-7
        for which in ('det', 'end'):
            assert (torch.tensor(state[which], dtype=torch.float64) - st['acc'][which]).abs().max().item() < 1e-12


# ---- the training-mode forward and its gradients against the IMPORTED reference's training step ----
from common import compare_train_step, load_train_case, train_case_names  # noqa: E402
from mmmot_amd.weights import generate_state_dict  # noqa: E402
from common import build_model  # noqa: E402


def test_train_fixtures_exist():
    names = train_case_names()
    assert len(names) >= 3 and any('3frames' in n for n in names) and any(n.endswith('_C') for n in names)


@pytest.mark.parametrize('name', train_case_names())
def test_oracle_training_step_matches_the_reference_fixture(name):
    """tracking_forward_train (batch-statistics BatchNorm in the trunk and w_det, unpadded new / end) + tracking_loss,
    differentiated by autograd, on the fixture's sample: outputs, loss, every gradient's norm, element-wise slices, and the
    BatchNorm buffers after the forward - all from the reference's own training step (tracking_model.py:50-66)."""
    c, kw, g, (dets, info, ds), gts = load_train_case(name)
    counts = [int(d) for d in ds]
    base = dict(c, fusion=c['fusion'], aff=c['aff'], sm=c['sm'])
    from common import manifest
    m = build_model(base, manifest()['base_kwargs'])  # the mirror module: only its state_dict (generated weights) is used
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith('idt')
              and 'running_' not in k else v.detach().clone()) for k, v in m.state_dict().items()}
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
    stats = {}
    if 'rng_seed' in c:  # DropBlock fixtures: the seed masks come from the global host generator, like the reference's
        torch.manual_seed(c['rng_seed'])
    det, links, new, end, trans = R.tracking_forward_train(sd, cfg, None, info['points'], info['points_split'], counts,
                                                           crops=dets, bn_stats=stats, dropblock=c.get('dropblock', 0))
    loss = R.tracking_loss(counts, gts[0], gts[1], gts[2], gts[3], det, links, new, end, trans, **kw)
    loss.backward()
    worst = compare_train_step(g, (det, links, new, end, trans), loss.item(),
                               lambda k: sd[k].grad if sd[k].requires_grad else None, stats,
                               out_tol=5e-5, loss_tol=2e-6, grad_tol=1e-3, norm_tol=1e-3, bn_tol=1e-5, what=name)
    print('%s: oracle vs the reference training step: %s' % (name, ' '.join('%s=%.1e' % kv for kv in worst.items())))
