"""Shared helpers for the test-suite (golden fixtures, model construction)."""
import json
import os

import numpy as np
import torch

from mmmot_amd import TrackingNet
from mmmot_amd.synth import make_pair
from mmmot_amd.weights import init_module, state_dict_for_profile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-3  # BASELINE.json north_star: outputs within 1e-3 (fp32) of the reference CPU path


def manifest():
    with open(os.path.join(GOLD, 'manifest.json')) as f:
        return json.load(f)


def case_names(light_only=False):
    """Small fixtures (outputs + stage checkpoints).  The full-size ones are in full_case_names()."""
    names = [c['name'] for c in manifest()['cases'] if not c.get('full')]
    if light_only:
        names = [n for n in names if not n.startswith('s3_')]
    return names


def full_case_names():
    """BASELINE.json-size fixtures (cfg3 / cfg4 shapes, outputs of the imported reference only)."""
    return [c['name'] for c in manifest()['cases'] if c.get('full')]


def get_case(name):
    m = manifest()
    c = [x for x in m['cases'] if x['name'] == name][0]
    return c, m['base_kwargs']


def case_inputs(c):
    if 'counts' in c:
        dets, info, _ = make_pair(c['counts'][0], sum(c['counts'][1:]), c['S'], c['pts'], c['seed'], ragged=True)
        return dets, info, [torch.tensor([x]) for x in c['counts']]
    return make_pair(c['N'], c['M'], c['S'], c['pts'], c['seed'], c['ragged'], reflectivity=bool(c.get('refl')))


def case_kwargs(c, base):
    kw = dict(base, score_fusion_arch=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'],
              end_mode=c.get('end_mode', base.get('end_mode', 'avg')))
    if c.get('refl'):
        kw['without_reflectivity'] = False
    if 'dropblock' in c:
        kw['dropblock'] = c['dropblock']
    if 'counts' in c:
        kw['seq_len'] = len(c['counts'])
    return kw


def build_model(c, base, device='cpu', ops=None):
    m = TrackingNet(**case_kwargs(c, base))
    if 'weights' in c:  # fixtures generated on other weight statistics than the seed-0 default (oracle/gen_golden.py)
        m.load_state_dict(state_dict_for_profile(m.state_dict(), c['weights'], c['S']), strict=True)
    else:
        init_module(m, 0)
    m.eval()
    if device != 'cpu':
        m = m.to(device)
    if ops is not None:
        m.set_ops(ops)
    return m


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def compare_outputs(out, g, tol=TOL, rows=(0, 1, 2)):
    """out = (det, links, new, end, trans) from a model/oracle; g = golden npz."""
    det, links, new, end, trans = out
    r = list(rows)
    errs = {
        'det': np.abs(det.detach().cpu().numpy() - g['det'][r]).max(),
        'new': np.abs(new.detach().cpu().numpy() - g['new'][r]).max(),
        'end': np.abs(end.detach().cpu().numpy() - g['end'][r]).max(),
        'link': max(np.abs(l.detach().cpu().numpy() - g['link%d' % i][r]).max() for i, l in enumerate(links)),
    }
    if trans is not None:
        errs['trans'] = max(np.abs(trans[0].detach().cpu().numpy() - g['trans1']).max(),
                            np.abs(trans[1].detach().cpu().numpy() - g['trans2']).max())
    bad = {k: float(v) for k, v in errs.items() if not (v <= tol)}
    assert not bad, 'outputs differ from the reference golden by more than %.0e: %r (all: %r)' % (tol, bad, errs)
    return errs


# ---- training-step fixtures of the IMPORTED reference (oracle/gen_golden_train.py) ----
def train_case_names():
    import glob
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, 'train_*.npz')))


def load_train_case(name):
    """-> (case dict, loss kwargs, fixture npz, inputs (dets, info, dets_split), gts (det, [link], new, end))"""
    import ast
    g = np.load(os.path.join(GOLD, name + '.npz'))
    c = dict(ast.literal_eval(str(g['case'])))
    kw = dict(ast.literal_eval(str(g['loss_kwargs'])))
    if 'counts' in c:
        dets, info, _ = make_pair(c['counts'][0], sum(c['counts'][1:]), c['S'], c['pts'], c['seed'], ragged=True)
        ds = [torch.tensor([x]) for x in c['counts']]
    else:
        dets, info, ds = make_pair(c['N'], c['M'], c['S'], c['pts'], c['seed'], True)
    n = len(ds) - 1
    t = lambda k: torch.from_numpy(g[k])
    gts = (t('gt_det'), [t('gt_link%d' % i) for i in range(n)], t('gt_new'), t('gt_end'))
    return c, kw, g, (dets, info, ds), gts


def compare_train_step(g, outs, loss, grad_of, buffers, out_tol, loss_tol, grad_tol, norm_tol, bn_tol, what):
    """One training step against a reference fixture.  outs = (det, [links], new, end, trans); grad_of(key) -> gradient
    tensor (or None = zero); buffers: key -> tensor for the 'bn:' entries.  Gradients are compared relative to the
    largest entry of the REFERENCE's gradient tensor (grad_norms[:, 2]); tensors whose reference gradient is rounding
    residue (< 1e-5: biases in front of a normalisation, the STN layers behind the one-value-per-group GroupNorm) must
    stay below 1e-4 absolutely.  Returns a dict of the worst differences."""
    det, links, new, end, trans = outs
    f = lambda x: x.detach().cpu().double().numpy()
    worst = dict(det=np.abs(f(det) - g['det']).max(), new=np.abs(f(new) - g['new']).max(), end=np.abs(f(end) - g['end']).max(),
                 link=max(np.abs(f(l) - g['link%d' % i]).max() for i, l in enumerate(links)),
                 trans=max(np.abs(f(trans[0]) - g['trans1']).max(), np.abs(f(trans[1]) - g['trans2']).max()),
                 loss=abs(float(loss) - float(g['loss'])))
    assert max(worst[k] for k in ('det', 'new', 'end', 'link', 'trans')) < out_tol, (what, worst)
    assert worst['loss'] < loss_tol * max(1.0, abs(float(g['loss']))), (what, worst)
    keys = [str(k) for k in g['grad_keys']]
    norms = {k: g['grad_norms'][i] for i, k in enumerate(keys)}
    sl, nm, res = 0.0, 0.0, 0.0
    for name in g.files:
        if not name.startswith('g:'):
            continue
        k = name[2:]
        want = g[name]
        got = grad_of(k)
        got = np.zeros_like(want) if got is None else f(got)[:want.shape[0]]
        sl = max(sl, np.abs(got - want).max() / norms[k][2])
    for k in keys:
        got = grad_of(k)
        got = None if got is None else f(got)
        s1, s2, amax = norms[k]
        if amax < 1e-5:
            if got is not None:
                res = max(res, np.abs(got).max())
            continue
        assert got is not None, (what, k, 'no gradient')
        nm = max(nm, abs(np.sqrt((got ** 2).sum()) - np.sqrt(s2)) / np.sqrt(s2), abs(np.abs(got).max() - amax) / amax)
    worst.update(grad_slices=sl, grad_norms=nm, grad_residue=res)
    assert sl < grad_tol and nm < norm_tol and res < 1e-4, (what, worst)
    bn = 0.0
    for name in g.files:
        if name.startswith('bn:'):
            got = buffers[name[3:]].detach().cpu().double().numpy()
            bn = max(bn, np.abs(got - g[name]).max())
    worst['bn_buffers'] = bn
    assert bn < bn_tol, (what, worst)
    return worst
