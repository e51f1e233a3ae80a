"""ORACLE tooling - pins the TrackingLoss restatement (oracle/restatement.py::tracking_loss) to the REAL reference.

Runs only in the build container (needs /root/reference).  Imports the reference's ``cost.py`` (it imports nothing
but torch) with ONE harness-side shim, no edits to the reference: ``cost.py:119,123`` compute ``1 - gt.eq(ignore)``,
legal on the torch 1.0 the reference targeted (``eq`` returned uint8) and a TypeError on torch >= 1.2 (bool); while the
reference's loss runs, ``Tensor.eq`` returns uint8 again.  For every case the reference's loss value and, through
``loss.backward()``, its gradients with respect to the four score tensors and the transforms are stored as
``tests/golden/loss_*.npz`` together with the inputs; the restatement is asserted against them here and again in
``tests/test_train_oracle.py`` wherever the repo runs.

    python oracle/gen_golden_loss.py
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402

REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')

CASES = [
    # name, frame counts, kwargs of TrackingLoss (the pp_pv_40e_mul_C config: det bce, link l2, det_ratio 1.5, trans 0.001)
    ('loss_cfg_mul_C', [6, 5], dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)),
    ('loss_defaults', [3, 7], dict(linkloss_type='l2')),  # the default 'l2_softmax' trips cost.py:73's assert
    ('loss_l1_l2det', [4, 4], dict(detloss_type='l2', endloss_type='l1', linkloss_type='l1', det_ratio=0.7, trans_ratio=0.4,
                                   trans_last=True)),
    ('loss_3frames', [3, 2, 4], dict(detloss_type='bce', linkloss_type='l2', det_ratio=0.4, trans_ratio=0.01)),
]


@contextlib.contextmanager
def uint8_eq():
    orig = torch.Tensor.eq
    torch.Tensor.eq = lambda self, other: orig(self, other).to(torch.uint8)
    try:
        yield
    finally:
        torch.Tensor.eq = orig


def make_inputs(counts, seed):
    g = torch.Generator().manual_seed(seed)
    L = sum(counts)
    rnd = lambda *s: torch.randn(*s, generator=g)
    det = rnd(3, L) * 2.0                                   # raw training-mode det scores (logits)
    links = [torch.rand(3, counts[i], counts[i + 1], generator=g) for i in range(len(counts) - 1)]
    new = torch.rand(3, L - counts[0], generator=g)          # sigmoid outputs, frames 1..
    end = torch.rand(3, L - counts[-1], generator=g)         # frames 0..-2
    gt_det = (torch.rand(L, generator=g) > 0.3).float()
    gt_new = (torch.rand(L, generator=g) > 0.6).float()
    gt_end = (torch.rand(L, generator=g) > 0.6).float()
    gt_new[torch.rand(L, generator=g) > 0.8] = -1.0          # ignore_index entries (DetLoss 'l2' / 'l1' mask)
    gt_end[torch.rand(L, generator=g) > 0.8] = -1.0
    gt_link = [(torch.rand(1, counts[i], counts[i + 1], generator=g) > 0.8).float() for i in range(len(counts) - 1)]
    trans = [torch.eye(3).unsqueeze(0) + 0.1 * rnd(1, 3, 3), torch.eye(64).unsqueeze(0) + 0.05 * rnd(1, 64, 64)]
    return det, links, new, end, gt_det, gt_link, gt_new, gt_end, trans


def main():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import cost as ref_cost
    worst = 0.0
    for k, (name, counts, kw) in enumerate(CASES):
        det, links, new, end, gt_det, gt_link, gt_new, gt_end, trans = make_inputs(counts, 500 + k)
        leaves = [t.clone().requires_grad_(True) for t in [det, new, end] + links + trans]
        d, n, e = leaves[:3]
        lk, tr = leaves[3:3 + len(links)], leaves[3 + len(links):]
        split = [torch.tensor(c) for c in counts]
        with contextlib.redirect_stdout(io.StringIO()):
            crit = ref_cost.TrackingLoss(**kw)
        with uint8_eq():
            loss = crit(split, gt_det, gt_link, gt_new, gt_end, d, lk, n, e, tr)
        loss.backward()
        # the restatement on the same inputs, float32 like the reference
        leaves2 = [t.clone().requires_grad_(True) for t in [det, new, end] + links + trans]
        loss2 = R.tracking_loss(counts, gt_det, gt_link, gt_new, gt_end, leaves2[0], leaves2[3:3 + len(links)], leaves2[1],
                                leaves2[2], leaves2[3 + len(links):], **kw)
        loss2.backward()
        err = abs(loss.item() - loss2.item())
        for a, b in zip(leaves, leaves2):
            ga = a.grad if a.grad is not None else torch.zeros_like(a)
            gb = b.grad if b.grad is not None else torch.zeros_like(b)
            err = max(err, (ga - gb).abs().max().item())
        worst = max(worst, err)
        assert err < 1e-6, (name, err)
        out = dict(counts=np.asarray(counts), loss=np.float32(loss.item()), det=det.numpy(), new=new.numpy(), end=end.numpy(),
                   gt_det=gt_det.numpy(), gt_new=gt_new.numpy(), gt_end=gt_end.numpy(),
                   g_det=d.grad.numpy(), g_new=n.grad.numpy(), g_end=e.grad.numpy())
        for i in range(len(links)):
            out['link%d' % i], out['gt_link%d' % i], out['g_link%d' % i] = links[i].numpy(), gt_link[i].numpy(), lk[i].grad.numpy()
        for i in range(2):
            out['trans%d' % i] = trans[i].numpy()
            out['g_trans%d' % i] = (tr[i].grad if tr[i].grad is not None else torch.zeros_like(tr[i])).numpy()
        out['kwargs'] = np.asarray(repr(sorted(kw.items())))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
        print('%-16s frames %-10s loss %.6f   |reference - restatement| <= %.1e' % (name, counts, loss.item(), err))
    print('worst |reference - restatement| over loss values and gradients: %.2e' % worst)
    ghm_sequence(ref_cost)


def ghm_sequence(ref_cost):
    """detloss_type / endloss_type 'ghm' (cost.py:105-110,126-128 -> modules/ghm_loss.py GHMC_Loss(bins=30, momentum=0.75)):
    the loss keeps running per-bin counts between calls, so the fixture is a SEQUENCE - one criterion, three consecutive
    samples; stored per step: inputs, loss, gradients and the running counts of both GHMC_Loss instances afterwards."""
    kw = dict(detloss_type='ghm', endloss_type='ghm', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
    # DetLoss('ghm') imports modules.ghm_loss, i.e. the `modules` package, whose __init__ imports torchvision (absent in
    # this image, only dereferenced on resnet paths): the empty stub of oracle/gen_golden.py - a harness shim, no edit
    import types
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tv.models)
    with contextlib.redirect_stdout(io.StringIO()):
        crit = ref_cost.TrackingLoss(**kw)
    state = dict(det=R.new_ghm_state(), end=R.new_ghm_state())
    out = dict(kwargs=np.asarray(repr(sorted(kw.items()))), steps=np.asarray(3))
    worst = 0.0
    for k, counts in enumerate([[6, 5], [4, 7], [6, 5]]):
        det, links, new, end, gt_det, gt_link, gt_new, gt_end, trans = make_inputs(counts, 700 + k)
        gt_det[torch.rand(sum(counts), generator=torch.Generator().manual_seed(710 + k)) > 0.85] = -1.0  # ignored detections
        leaves = [t.clone().requires_grad_(True) for t in [det, new, end] + links + trans]
        split = [torch.tensor(c) for c in counts]
        with uint8_eq():
            loss = crit(split, gt_det, gt_link, gt_new, gt_end, leaves[0], leaves[3:4], leaves[1], leaves[2], leaves[4:])
        loss.backward()
        leaves2 = [t.clone().requires_grad_(True) for t in [det, new, end] + links + trans]
        loss2 = R.tracking_loss(counts, gt_det, gt_link, gt_new, gt_end, leaves2[0], leaves2[3:4], leaves2[1], leaves2[2],
                                leaves2[4:], ghm_state=state, **kw)
        loss2.backward()
        err = abs(loss.item() - loss2.item())
        for a, b in zip(leaves, leaves2):
            ga = a.grad if a.grad is not None else torch.zeros_like(a)
            gb = b.grad if b.grad is not None else torch.zeros_like(b)
            err = max(err, (ga - gb).abs().max().item())
        acc_det, acc_end = crit.det_loss.GHMC_Loss.acc_sum, crit.end_loss.GHMC_Loss.acc_sum
        err = max(err, max(abs(a - b) for a, b in zip(acc_det + acc_end, state['det'] + state['end'])))
        worst = max(worst, err)
        assert err < 1e-6, ('lossseq_ghm', k, err)
        t = '_%d' % k
        out.update({'counts' + t: np.asarray(counts), 'loss' + t: np.float32(loss.item()), 'det' + t: det.numpy(),
                    'new' + t: new.numpy(), 'end' + t: end.numpy(), 'gt_det' + t: gt_det.numpy(), 'gt_new' + t: gt_new.numpy(),
                    'gt_end' + t: gt_end.numpy(), 'link0' + t: links[0].numpy(), 'gt_link0' + t: gt_link[0].numpy(),
                    'trans0' + t: trans[0].numpy(), 'trans1' + t: trans[1].numpy(),
                    'g_det' + t: leaves[0].grad.numpy(), 'g_new' + t: leaves[1].grad.numpy(), 'g_end' + t: leaves[2].grad.numpy(),
                    'g_link0' + t: leaves[3].grad.numpy(),
                    'acc_det' + t: np.asarray(acc_det, np.float64), 'acc_end' + t: np.asarray(acc_end, np.float64)})
        print('lossseq_ghm step %d frames %-8s loss %.6f   |reference - restatement| <= %.1e' % (k, counts, loss.item(), err))
    np.savez_compressed(os.path.join(GOLD, 'lossseq_ghm.npz'), **out)


if __name__ == '__main__':
    main()
