"""Shared helpers for the test-suite (golden fixtures, model construction)."""
import json
import os

import numpy as np
import torch

from mmmot_amd import TrackingNet
from mmmot_amd.synth import make_pair
from mmmot_amd.weights import init_module

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
    if 'counts' in c:
        kw['seq_len'] = len(c['counts'])
    return kw


def build_model(c, base, device='cpu', ops=None):
    m = TrackingNet(**case_kwargs(c, base))
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
