"""The oracle (oracle/restatement.py) against the golden vectors that
oracle/gen_golden.py recorded from the REAL reference (imported from
/root/reference in the build container).  This is what pins the oracle."""
import pytest
import torch

from common import case_inputs, case_names, compare_outputs, full_case_names, get_case, golden
from mmmot_amd.weights import state_dict_for_profile
from mmmot_amd import TrackingNet
from common import case_kwargs
from oracle import restatement as R

_SD = {}


def state_dict_for(c, base):
    key = (c['fusion'], len(c.get('counts', [0, 0])), bool(c.get('refl')), c.get('weights', 'default:0'),
           c['S'] if 'weights' in c else 0)
    if key not in _SD:
        spec = TrackingNet(**case_kwargs(c, base)).state_dict()
        _SD[key] = state_dict_for_profile(spec, key[3], c['S'])
    return _SD[key]


@pytest.mark.parametrize('name', case_names() + full_case_names())
def test_oracle_matches_reference_golden(name):
    c, base = get_case(name)
    sd = state_dict_for(c, base)
    dets, info, ds = case_inputs(c)
    cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'], neg_threshold=base['neg_threshold'],
               score_arch=base['score_arch'], end_mode=c.get('end_mode', 'avg'))
    with torch.no_grad():
        out = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], [int(d) for d in ds])
    errs = compare_outputs(out, golden(name), tol=5e-5)
    assert errs['trans'] < 1e-6


def test_oracle_single_modality_rows_equal_full_rows():
    """Modality rows never mix: evaluating row r alone gives the golden row r."""
    c, base = get_case('s2_C_multiply_none')
    sd = state_dict_for(c, base)
    dets, info, ds = case_inputs(c)
    cfg = dict(fusion='C', affinity_op=c['aff'], softmax_mode=c['sm'], neg_threshold=base['neg_threshold'])
    g = golden(c['name'])
    for rows in ((0,), (1,)):
        with torch.no_grad():
            out = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], [int(d) for d in ds], rows=rows)
        out = out[:4] + (None,)
        compare_outputs(out, g, tol=5e-5, rows=rows)
