"""ORACLE tooling - pins the TRAINING-mode restatement (oracle/restatement.py::tracking_forward_train + tracking_loss,
differentiated by torch.autograd) to the REAL reference's training step.

Runs only in the build container (needs /root/reference).  The reference's training step is
``tracking_model.py:50-66``: ``model.train()`` forward -> ``TrackingLoss`` (``cost.py:134-185``) -> ``loss.backward()``.
Here the imported ``modules.TrackingNet(...).train()`` and the imported ``cost.TrackingLoss`` run that step on small
samples (same three import shims as oracle/gen_golden.py, plus the ``Tensor.eq -> uint8`` shim of
oracle/gen_golden_loss.py while the loss runs; no edits to the reference) and the script stores, per case,

* the training-mode outputs (raw det scores, link scores, UNPADDED new / end scores, transforms) and the loss;
* of EVERY parameter's gradient: sum, sum of squares and absolute maximum (three numbers per tensor);
* element-wise gradients of one tensor per block (slices where the tensor is large): the first VGG conv and its
  BatchNorm, a deep VGG conv, a SkipPool layer, PointNet conv5 / conv1, an STN output layer, the fusion block, w_det, the
  pairwise block (w_link.conv1.0, w_new_end.conv0.0);
* the BatchNorm buffers (running_mean / running_var / num_batches_tracked) of the trunk and of w_det AFTER the forward.

``tests/test_train_oracle.py`` asserts the restatement against these fixtures on the CPU and ``tests/test_train_vgg_gpu.py``
the device step (``-m gpu``).  The fixtures: ``tests/golden/train_*.npz``.

    python oracle/gen_golden_train.py
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
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from gen_golden import BASE, REF, GOLD, import_reference, make_multiframe  # noqa: E402
from gen_golden_loss import uint8_eq  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import generate_state_dict  # noqa: E402
from oracle import restatement as R  # noqa: E402

LOSS_KW = dict(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)  # experiments/pp_pv_40e_mul_C

CASES = [
    dict(name='train_s2_C', fusion='C', aff='multiply', sm='none', N=5, M=7, S=64, pts=40, seed=1002, gt_seed=41),
    dict(name='train_s2_A_subabs', fusion='A', aff='minus_abs', sm='dual_add', N=6, M=5, S=64, pts=60, seed=1021, gt_seed=42),
    dict(name='train_s5_3frames_B', fusion='B', aff='multiply', sm='dual_add', counts=[3, 4, 2], S=32, pts=20, seed=1005,
         gt_seed=43),
    # DropBlock in the SkipPools of stages 2 / 3 (appear_net.py:17-18,28-29,143-152; modules/dropblock.py): block size 5,
    # drop probability 0.1; the seed masks are drawn on the HOST with the global torch generator - `rng_seed` is set right
    # before the forward, so whoever draws in the same order (stage 2, then stage 3) sees the same masks
    dict(name='train_s9_dropblock_C', fusion='C', aff='multiply', sm='none', N=7, M=6, S=96, pts=30, seed=1031, gt_seed=44,
         dropblock=5, rng_seed=61),
]

# element-wise gradient checks: key -> slice of the leading dimension (None = the whole tensor)
GRAD_SLICES = {
    'appearance.layers.0.0.weight': None, 'appearance.layers.0.1.weight': None, 'appearance.layers.0.1.bias': None,
    'appearance.layers.2.3.weight': 4, 'appearance.layers.3.7.weight': None,
    'appearance.global_pool.1.fc.1.weight': 8,
    'point_net.feat.conv1.weight': None, 'point_net.feat.conv5.weight': 32, 'point_net.conv1.weight': 8,
    'point_net.feat.stn1.output.weight': None, 'point_net.feat.stn2.output.weight': 32,
    'point_net.feat.stn2.fc_bn2.bias': None, 'point_net.bn2.weight': None,
    'w_link.conv1.0.weight': 16, 'w_link.conv1.9.weight': None, 'w_link.w_new_end.conv0.0.weight': 16,
    'w_link.w_new_end.conv1.6.weight': None, 'w_det.0.weight': 16, 'w_det.1.weight': None, 'w_det.6.weight': None,
}
FUSION_SLICES = {'A': {'fusion_module.input_w.0.weight': 16, 'fusion_module.input_w.1.bias': None},
                 'B': {'fusion_module.input_p.0.weight': 16, 'fusion_module.input_i.1.weight': None},
                 'C': {'fusion_module.gate_p.0.weight': 16, 'fusion_module.input_i.0.weight': 16,
                       'fusion_module.gate_i.0.bias': None}}


def make_gts(counts, seed):
    """ground-truth vectors of one sample (tests/test_train_cpu.py::make_gts draws the same way)"""
    g = torch.Generator().manual_seed(seed)
    L = sum(counts)
    gt_det = (torch.rand(L, generator=g) > 0.3).float()
    gt_new, gt_end = (torch.rand(L, generator=g) > 0.6).float(), (torch.rand(L, generator=g) > 0.6).float()
    gt_link = [(torch.rand(1, counts[i], counts[i + 1], generator=g) > 0.8).float() for i in range(len(counts) - 1)]
    return gt_det, gt_link, gt_new, gt_end


def case_inputs(c):
    if 'counts' in c:
        return make_multiframe(c['counts'], c['S'], c['pts'], c['seed'])
    return make_pair(c['N'], c['M'], c['S'], c['pts'], c['seed'], True)


def grad_slices(c):
    return dict(GRAD_SLICES, **FUSION_SLICES[c['fusion']])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    ref_modules = import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        import cost as ref_cost
    worst = {}
    for c in CASES:
        counts = c.get('counts', [c.get('N'), c.get('M')])
        kw = dict(BASE, score_fusion_arch=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'], seq_len=len(counts),
                  dropblock=c.get('dropblock', 0))
        with contextlib.redirect_stdout(io.StringIO()):
            model = ref_modules.TrackingNet(**kw)
            crit = ref_cost.TrackingLoss(**LOSS_KW)
        sd0 = generate_state_dict(model.state_dict(), seed=0)
        model.load_state_dict(sd0, strict=True)
        model.train()
        dets, info, dsplit = case_inputs(c)
        gts = make_gts(counts, c['gt_seed'])
        if 'rng_seed' in c:
            torch.manual_seed(c['rng_seed'])
            if c.get('dropblock'):  # the fixture must exercise the layer: at least one dropped block in each of the two stages
                probe = [float((torch.rand(sum(counts), c['S'] // d, c['S'] // d) < 0.1 / c['dropblock'] ** 2).sum()) for d in (16, 32)]
                assert min(probe) >= 1, ('rng_seed %d drops nothing in one stage: %r' % (c['rng_seed'], probe))
            torch.manual_seed(c['rng_seed'])
        det, links, new, end, trans = model(dets, info, dsplit)
        with uint8_eq():
            loss = crit(dsplit, gts[0], gts[1], gts[2], gts[3], det, links, new, end, trans)
        loss.backward()
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
        after = {k: v.detach().clone() for k, v in model.state_dict().items()}

        # ---- the restatement on the same sample (float32 like the reference), through autograd ----
        sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k in grads and k.split('.')[-1] != 'idt'
                  else v.clone()) for k, v in sd0.items()}
        cfg = dict(fusion=c['fusion'], affinity_op=c['aff'], softmax_mode=c['sm'])
        stats = {}
        if 'rng_seed' in c:
            torch.manual_seed(c['rng_seed'])
        o_det, o_links, o_new, o_end, o_trans = R.tracking_forward_train(
            sd, cfg, None, info['points'], info['points_split'], [int(d) for d in dsplit], crops=dets, bn_stats=stats,
            dropblock=c.get('dropblock', 0))
        o_loss = R.tracking_loss(counts, gts[0], gts[1], gts[2], gts[3], o_det, o_links, o_new, o_end, o_trans, **LOSS_KW)
        o_loss.backward()
        e = dict(det=(det - o_det).abs().max().item(), new=(new - o_new).abs().max().item(),
                 end=(end - o_end).abs().max().item(), loss=abs(loss.item() - o_loss.item()),
                 link=max((a - b).abs().max().item() for a, b in zip(links, o_links)))
        # gradients: relative to the tensor's largest entry, for tensors whose gradient is not rounding residue (biases in
        # front of a normalisation, the STN layers behind the one-value-per-group GroupNorm: mathematically zero, 1e-7 in
        # the reference's fp32) - those are compared absolutely
        rel, resid = 0.0, 0.0
        for k, g in grads.items():
            og = sd[k].grad if (sd[k].requires_grad and sd[k].grad is not None) else torch.zeros_like(g)
            d = (g - og).abs().max().item()
            if g.abs().max().item() > 1e-5:
                rel = max(rel, d / g.abs().max().item())
            else:
                resid = max(resid, d)
        e['grad_rel'], e['grad_residue_abs'] = rel, resid
        e['bn'] = max((after[k] - v).abs().max().item() for k, v in stats.items())
        worst[c['name']] = e
        print('%-22s loss %.6f  |reference - restatement|: %s' % (c['name'], loss.item(),
              ' '.join('%s=%.1e' % kv for kv in e.items())), flush=True)
        assert max(e['det'], e['new'], e['end'], e['link'], e['loss']) < 5e-5 and e['bn'] < 1e-5, e
        assert e['grad_rel'] < 2e-3 and e['grad_residue_abs'] < 1e-5, e

        # ---- the fixture ----
        out = dict(counts=np.asarray(counts), loss=np.float32(loss.item()), det=det.detach().numpy(),
                   new=new.detach().numpy(), end=end.detach().numpy(), trans1=trans[0].detach().numpy(),
                   trans2=trans[1].detach().numpy(), gt_det=gts[0].numpy(), gt_new=gts[2].numpy(), gt_end=gts[3].numpy())
        for i, l in enumerate(links):
            out['link%d' % i], out['gt_link%d' % i] = l.detach().numpy(), gts[1][i].numpy()
        keys = sorted(grads)
        out['grad_keys'] = np.asarray(keys)
        out['grad_norms'] = np.asarray([[grads[k].double().sum().item(), (grads[k].double() ** 2).sum().item(),
                                         grads[k].abs().max().item()] for k in keys], np.float64)
        for k, n in grad_slices(c).items():
            g = grads[k]
            out['g:' + k] = (g if n is None else g[:n]).contiguous().numpy()
        for k, v in after.items():
            if 'running_' in k or 'num_batches_tracked' in k:
                out['bn:' + k] = v.numpy()
        out['case'] = np.asarray(repr(sorted(c.items())))
        out['loss_kwargs'] = np.asarray(repr(sorted(LOSS_KW.items())))
        np.savez_compressed(os.path.join(GOLD, c['name'] + '.npz'), **out)
    print('fixtures written: %s' % ', '.join(c['name'] for c in CASES))


if __name__ == '__main__':
    main()
