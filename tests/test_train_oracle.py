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
