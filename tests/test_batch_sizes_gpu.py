"""BASELINE.json configurations at their BATCH sizes through one ``forward_batch`` (the suite ran cfg3 with
B = 2 and small shapes with B = 3 only): cfg2 = 32 frame pairs of N = M = 32, cfg4 = 32 pairs per GPU of N = M = 128
(256 pairs over 8 GPUs), cfg3 = the 16 pairs per step bench.py times.  Checked: the first pair against the output of
the IMPORTED reference (golden, same seed), first / last pair bitwise equal to the same pair run alone (no statistic,
tile or table crosses a sample), and the whole batch finite."""
import numpy as np
import pytest
import torch

from common import TOL, build_model, get_case, golden
from mmmot_amd.synth import make_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def run_batch(m, c, B, seed0):
    ins = [make_pair(c['N'], c['M'], c['S'], c['pts'], seed=seed0 + i) for i in range(B)]
    samples = [([c['N'], c['M']], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins]
    plan = m.make_plan(samples, c['S'])
    crops = torch.cat([x[0] for x in ins]).to(DEV)
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).to(DEV)
    with torch.no_grad():
        res = m.forward_batch(plan, crops, points)
    res = [(d.clone(), [l.clone() for l in ls], n.clone(), e.clone()) for d, ls, n, e in res]
    return ins, res


def single(m, x):
    dets, info, ds = x
    with torch.no_grad():
        det, links, new, end, _ = m(dets.to(DEV), {k: v.to(DEV) for k, v in info.items()}, ds)
    return det.clone(), [l.clone() for l in links], new.clone(), end.clone()


@pytest.mark.parametrize('name,B', [('s4_cfg2_A', 32), ('f_cfg4_C', 32), ('f_cfg3_C', 16)])
def test_config_batch_size_in_one_forward_batch(name, B):
    c, base = get_case(name)
    m = build_model(c, base, device=DEV)   # default arithmetic (f16x3)
    ins, res = run_batch(m, c, B, c['seed'])
    assert len(res) == B
    for det, links, new, end in res:
        assert all(torch.isfinite(t).all() for t in (det, links[0], new, end))
    # first pair = the golden's seed: against the imported reference
    g = golden(name)
    det, links, new, end = res[0]
    err = max(np.abs(det.cpu().numpy() - g['det']).max(), np.abs(links[0].cpu().numpy() - g['link0']).max(),
              np.abs(new.cpu().numpy() - g['new']).max(), np.abs(end.cpu().numpy() - g['end']).max())
    print(name, 'B=%d' % B, 'pair 0 vs reference golden: %.2e' % err)
    assert err < TOL
    # batched == alone, bit for bit (first and last sample of the batch)
    for i in (0, B - 1):
        sd, sl, sn, se = single(m, ins[i])
        bd, bl, bn, be = res[i]
        assert torch.equal(sd, bd) and torch.equal(sl[0], bl[0]) and torch.equal(sn, bn) and torch.equal(se, be), \
            'sample %d of the batch differs from the same sample run alone' % i


def test_captured_forward_follows_refresh_head_and_goes_stale_on_repack():
    """a hipGraph holds raw pointers into the packed head.  refresh_head() copies in place - the next
    replay computes with the new weights; a re-pack that replaces tensors makes the graph raise instead of replaying."""
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, device=DEV)
    x = make_pair(c['N'], c['M'], c['S'], c['pts'], seed=c['seed'], ragged=c['ragged'])
    plan = m.make_plan([([c['N'], c['M']], x[1]['points_split'].reshape(-1).long().numpy())], c['S'])
    crops, points = x[0].to(DEV), x[1]['points'].reshape(-1, 3).to(DEV)
    g = m.capture(plan, crops, points)
    a = [t.clone() for t in (g(crops, points)[0][0], g(crops, points)[0][1][0])]
    with torch.no_grad():
        m.w_link.conv1[3].weight.mul_(1.01)
        m.w_det[0].weight.mul_(0.99)
    assert not m.head_is_current()
    m.refresh_head()
    if not g.stale():   # scalars (final biases, hl16 scales) unchanged: the in-place path
        b = g(crops, points)[0]
        eager = m.forward_batch(plan, crops, points)[0]
        assert torch.equal(b[0], eager[0]) and torch.equal(b[1][0], eager[1][0])
        assert not torch.equal(b[1][0], a[1])
    assert g.check_range() == (0, 0, 0)
    m.invalidate()
    with pytest.raises(RuntimeError, match='stale'):
        g(crops, points)
