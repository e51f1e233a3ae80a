"""Training step, second slice: LiDAR encoder backward, training-mode forward, TrackingLoss.

The reference's training step is ``tracking_model.py:50-66``: training-mode ``TrackingNet.forward`` ->
``generate_gt`` (host) -> ``TrackingLoss`` (``cost.py:134-185``) -> ``loss.backward()`` -> ``optimizer.step()``.
``mmmot_amd/backward.py`` built the head (fusion, w_det in training mode, the pairwise block); this file adds

* ``pointnet_autograd(model, plan, points)``: ``PointNet_v1.forward`` (reference modules/point_net.py:25-44, 115-153)
  as a differentiable operator.  PointNet contains GroupNorm only (no BatchNorm; dropout is off in every config), so its
  training forward is the inference arithmetic - run here in its MATERIALISING form (the pre-norm tensors
  of every layer are the tape; the inference engine never stores the [P][1024] / [P][512] ones) and with exact-fp32
  matrix-core GEMMs on the unscaled weights (no per-step hl16 re-packing).  Backward per layer: the GroupNorm(+ReLU)
  backward, ``dW = dY^T A`` and ``dA = dY W`` kernels of csrc/backward.hip, plus the two kernels of csrc/train.hip
  (average-pool backward, first-layer weight gradient).
  The two spatial transforms are input-independent (STN3d ends in GroupNorm(C, C) over ONE value per group:
  ``trans = output(relu(fc_bn2.bias)) + I``, and every other STN parameter gets an exactly zero gradient), so they are
  folded into the adjacent 1x1 convs as at inference; here the fold is done with torch operations on the 64 x 64
  matrices UNDER AUTOGRAD, which is what carries the gradients of the folded weights back to ``conv*.weight``,
  ``stn*.output.*`` and ``stn*.fc_bn2.bias`` (and lets the loss' transform regulariser act on ``trans``).
* ``TrackingLoss``: the reference's criterion with the reference's signature; the score terms and their gradients come
  from ``mmmot_score_loss`` in one pass, the transform regulariser (a 64 x 64 expression) stays in torch.
* ``forward_train(model, dets, det_info, dets_split)``: the training-mode forward with the reference's return contract
  (raw det scores, no eval padding of new / end), attached to autograd: image encoder in training mode
  (mmmot_amd/train_vgg.py: batch-statistics BatchNorm2d, differentiable) unless ``model.freeze_appearance`` is set, in
  which case the image branch runs frozen on the eval-mode inference trunk (folded running statistics, no gradient).
"""
import collections
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as Fn

from .backward import _aux, _colsum, _current_engine, _gn_backward, _norm_layer, dgrad_gemm, head_autograd
from .ops import ACT_RELU, A_NORM_RELU, A_PLAIN, LOSS_KINDS
from .plan import Segments

# folded tensor name -> produced by fold_pointnet(); order = argument order of _PointNetFn
FOLDED = ('w1', 'b1', 'g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3', 'g3', 'be3', 'w4', 'b4', 'g4', 'be4',
          'w5', 'b5', 'g5', 'be5', 'wc1a', 'wc1b', 'bc1', 'gc1', 'bec1', 'wc2', 'bc2', 'gc2', 'bec2')


def stn_transform(stn, k):
    """closed form of STN3d.forward (reference modules/point_net.py:72-86), differentiable"""
    h = torch.relu(stn.fc_bn2.bias)
    return Fn.linear(h, stn.output.weight, stn.output.bias).view(k, k) + stn.idt


def fold_pointnet(point_net):
    """The algebra of mmmot_amd.pack.pack_weights for PointNet, on the live parameters and differentiable:
    returns ({name: fp32 tensor}, [trans1 (1, k, k), trans2 (1, 64, 64)])."""
    f = point_net.feat
    kin = int(f.conv1.weight.shape[1])
    T1, T2 = stn_transform(f.stn1, kin), stn_transform(f.stn2, 64)
    cw = lambda conv: conv.weight.flatten(1)
    W = {'w1': cw(f.conv1) @ T1.t(), 'b1': f.conv1.bias, 'w2': cw(f.conv2) @ T2.t(), 'b2': f.conv2.bias}
    for i in (3, 4, 5):
        W['w%d' % i], W['b%d' % i] = cw(getattr(f, 'conv%d' % i)), getattr(f, 'conv%d' % i).bias
    for i in (1, 2, 3, 4, 5):
        W['g%d' % i], W['be%d' % i] = getattr(f, 'bn%d' % i).weight, getattr(f, 'bn%d' % i).bias
    wc1 = cw(point_net.conv1)
    W['wc1a'], W['wc1b'], W['bc1'] = wc1[:, :64] @ T2.t(), wc1[:, 64:], point_net.conv1.bias
    W['gc1'], W['bec1'] = point_net.bn1.weight, point_net.bn1.bias
    W['wc2'], W['bc2'] = cw(point_net.conv2), point_net.conv2.bias
    W['gc2'], W['bec2'] = point_net.bn2.weight, point_net.bn2.bias
    return W, [T1.unsqueeze(0), T2.unsqueeze(0)]


class _PointAux:
    """tables of the PointNet backward the forward plan does not carry"""

    def __init__(self, plan):
        cnt = np.diff(plan.pt_split)
        self.inv_cnt = torch.from_numpy((1.0 / cnt).astype(np.float32)).to(plan.device)
        det_sample = np.repeat(np.arange(plan.B), plan.L)
        # SUM of a detection's point rows (the gradient of a per-detection bias added to each of them)
        self.det_sum = Segments(plan.pt_split[:-1], cnt, np.ones(plan.Lt), det_sample, plan.device, div=np.ones(plan.Lt))


def _paux(plan):
    if not hasattr(plan, '_pn_bwd_aux'):
        plan._pn_bwd_aux = _PointAux(plan)
    return plan._pn_bwd_aux


def _weight_grad(eng, dY, tiles, N, K, **kw):
    """(dW [N][K], db [N]) = (dY^T A, column sums of dY); rows split into shares whose partials are added (deterministic)"""
    dev = dY.device
    ns = max(1, min(16, tiles.T // 8))
    dWp = torch.empty(ns, N * K, dtype=torch.float32, device=dev)
    dbp = torch.empty(ns, N, dtype=torch.float32, device=dev)
    eng.ops.gemm_tn(dY, tiles, N, K, dWp, dbp, nsplit=ns, **kw)
    if ns > 1:
        return _colsum(eng, dWp).view(N, K), _colsum(eng, dbp)
    return dWp.view(N, K), dbp.view(N)


def pointnet_forward_train(eng, plan, points, W):
    """points [P][3 | 4], W = folded fp32 tensors (fold_pointnet) -> (features [Lt][512], tape)."""
    ops, T, D, Pn, Lt = eng.ops, plan.pt_tiles, plan.det_tiles, plan.P, plan.Lt
    dev = points.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    t = {}
    y1, part = new(Pn, 64), new(T.T, 2, 64)
    ops.pointnet_layer1(points, W['w1'], W['b1'], y1, part, T)
    L = t['p1'] = _norm_layer(eng, part, T, y1, 64, 64, W['g1'], W['be1'])
    x = y1
    for i, (N, K) in zip((2, 3, 4, 5), ((64, 64), (64, 64), (128, 64), (1024, 128))):
        y, part = new(Pn, N), new(T.T, 2, N)
        ops.gemm(W['w%d' % i], T, N, K, X=x, bias=W['b%d' % i], Y=y, part=part, sc=L.sc, sh=L.sh, amode=A_NORM_RELU)
        L = t['p%d' % i] = _norm_layer(eng, part, T, y, N, N, W['g%d' % i], W['be%d' % i])
        x = y
    seg1024 = new(Lt, 1024)
    ops.segment_mean(x, 1024, plan.det_segs, seg1024, sc=L.sc, sh=L.sh, relu=True)      # point_net.py:139-146
    dbias = new(Lt, 512)
    ops.gemm(W['wc1b'], D, 512, 1024, X=seg1024, bias=W['bc1'], Y=dbias)                # the broadcast 1024 channels
    yc1, part = new(Pn, 512), new(T.T, 2, 512)
    ops.gemm(W['wc1a'], T, 512, 64, X=y1, Y=yc1, part=part, sc=t['p1'].sc, sh=t['p1'].sh, amode=A_NORM_RELU,
             dbias=dbias, rowidx=plan.row_det)
    t['c1'] = _norm_layer(eng, part, T, yc1, 512, 512, W['gc1'], W['bec1'])
    seg512 = new(Lt, 512)
    ops.segment_mean(yc1, 512, plan.det_segs, seg512, sc=t['c1'].sc, sh=t['c1'].sh, relu=True)  # point_net.py:32-39
    yc2, part = new(Lt, 512), new(D.T, 2, 512)
    ops.gemm(W['wc2'], D, 512, 512, X=seg512, bias=W['bc2'], Y=yc2, part=part)
    t['c2'] = _norm_layer(eng, part, D, yc2, 512, 16, W['gc2'], W['bec2'])
    out = new(Lt, 512)
    ops.affine_act(yc2, 512, t['c2'].sc, t['c2'].sh, D, ACT_RELU, out)
    t.update(seg1024=seg1024, seg512=seg512)
    return out, t


def pointnet_backward(eng, plan, points, W, t, dOut):
    """dOut [Lt][512] -> {folded name: gradient}"""
    ops, T, D, Pn, Lt = eng.ops, plan.pt_tiles, plan.det_tiles, plan.P, plan.Lt
    dev = points.device
    aux = _paux(plan)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    g = {}
    p1 = t['p1']
    # ---- PointNet_v1 head: conv2 + GroupNorm(16) + ReLU over the detections (point_net.py:40) ----
    dyc2, g['gc2'], g['bec2'] = _gn_backward(eng, plan, t['c2'], dOut)
    g['wc2'], g['bc2'] = _weight_grad(eng, dyc2, D, 512, 512, X=t['seg512'], amode=A_PLAIN)
    dseg512 = new(Lt, 512)
    dgrad_gemm(eng, W['wc2'], D, dyc2, dseg512)
    # ---- per-detection average of relu(gn(conv1)) (point_net.py:32-39), conv1 1088 -> 512 + GroupNorm(512) ----
    dAc1 = new(Pn, 512)
    ops.rows_gather_scale(dseg512, plan.row_det, aux.inv_cnt, dAc1, 512)
    dyc1, g['gc1'], g['bec1'] = _gn_backward(eng, plan, t['c1'], dAc1)
    g['wc1a'], _ = _weight_grad(eng, dyc1, T, 512, 64, X=p1.Y, sc=p1.sc, sh=p1.sh, amode=A_NORM_RELU)
    ddbias = new(Lt, 512)  # the 1024 broadcast channels act as a per-detection bias: its gradient is the row SUM
    ops.segment_mean(dyc1, 512, aux.det_sum, ddbias, use_group=False)
    g['wc1b'], g['bc1'] = _weight_grad(eng, ddbias, D, 512, 1024, X=t['seg1024'], amode=A_PLAIN)
    dseg1024 = new(Lt, 1024)
    dgrad_gemm(eng, W['wc1b'], D, ddbias, dseg1024)
    dA1c = new(Pn, 64)
    dgrad_gemm(eng, W['wc1a'], T, dyc1, dA1c)
    # ---- PointNetfeatGN: average of relu(gn5(conv5)) back to the points, then the conv5 .. conv2 chain ----
    dA = new(Pn, 1024)
    ops.rows_gather_scale(dseg1024, plan.row_det, aux.inv_cnt, dA, 1024)
    for i, (N, K) in zip((5, 4, 3, 2), ((1024, 128), (128, 64), (64, 64), (64, 64))):
        Li, Lp = t['p%d' % i], t['p%d' % (i - 1)]
        dy, g['g%d' % i], g['be%d' % i] = _gn_backward(eng, plan, Li, dA)
        g['w%d' % i], g['b%d' % i] = _weight_grad(eng, dy, T, N, K, X=Lp.Y, sc=Lp.sc, sh=Lp.sh, amode=A_NORM_RELU)
        dA = new(Pn, K)
        dgrad_gemm(eng, W['w%d' % i], T, dy, dA)
    dA1 = new(Pn, 64)
    ops.add_rows(dA, dA1c, dA1, 64)  # relu(gn1(.)) feeds conv2 and the 64-channel skip of PointNet_v1.conv1
    dy1, g['g1'], g['be1'] = _gn_backward(eng, plan, p1, dA1)
    kin = int(points.shape[1])
    PW = new(T.T, 64 * (kin + 1))
    ops.pointnet_layer1_bwd(dy1, points, T, PW)
    pw = _colsum(eng, PW).view(64, kin + 1)
    g['w1'], g['b1'] = pw[:, :kin], pw[:, kin]
    return g


class _PointNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, eng, plan, *folded):
        W = {k: v.detach().to(torch.float32).contiguous() for k, v in zip(FOLDED, folded)}
        pts = points.detach().contiguous()
        out, tape = pointnet_forward_train(eng, plan, pts, W)
        ctx.eng, ctx.plan, ctx.W, ctx.tape, ctx.pts = eng, plan, W, tape, pts
        ctx.shapes = [tuple(v.shape) for v in folded]
        return out

    @staticmethod
    def backward(ctx, d_out):
        g = pointnet_backward(ctx.eng, ctx.plan, ctx.pts, ctx.W, ctx.tape, d_out.contiguous())
        return (None, None, None) + tuple(g[k].reshape(s) for k, s in zip(FOLDED, ctx.shapes))


def pointnet_autograd(model, plan, points):
    """Differentiable ``model.point_net`` (a TrackingNet's PointNet_v1 on the device): points [P][3 | 4] ->
    (features [Lt][512] on the autograd graph of every ``point_net.*`` parameter, [trans1, trans2])."""
    eng = model.engine()
    W, trans = fold_pointnet(model.point_net)
    out = _PointNetFn.apply(points, eng, plan, *[W[k] for k in FOLDED])
    return out, trans


# ======================================================================================================================
# TrackingLoss (reference cost.py:134-185)
# ======================================================================================================================
class _ScoreLossFn(torch.autograd.Function):
    """sum of the score terms of TrackingLoss; the kernel that evaluates a term also writes its gradient"""

    @staticmethod
    def forward(ctx, ops, terms, *scores):
        dev = scores[0].device
        PL = torch.zeros(1, dtype=torch.float32, device=dev)
        grads = []
        for k, (si, x2d, y, kind, scale, mask) in enumerate(terms):
            x = scores[si].detach()
            x = x.reshape(x2d).contiguous() if x2d is not None else x.contiguous()
            g = torch.empty_like(x)
            if kind == 'ghm':  # mask['state']: the module's running per-bin counts (updated in place by the kernel)
                ops.ghm_loss(x, y, scale, g, PL, mask['state'], bins=mask['bins'], momentum=mask['momentum'],
                             accumulate=(k > 0))
            else:
                ops.score_loss(x, y, kind, scale, g, PL, accumulate=(k > 0), **mask)
            grads.append((si, g))
        ctx.grads, ctx.n = grads, len(scores)
        ctx.shapes = [tuple(s.shape) for s in scores]
        return PL[0]

    @staticmethod
    def backward(ctx, d_loss):
        out = [None] * ctx.n
        for si, g in ctx.grads:
            gi = (g * d_loss).reshape(ctx.shapes[si])
            out[si] = gi if out[si] is None else out[si] + gi
        return (None, None) + tuple(out)


class TrackingLoss(nn.Module):
    """reference cost.py:134-185, same constructor and call signature.  ``detloss_type`` / ``endloss_type``: 'bce',
    'l2', 'l1' (DetLoss, cost.py:97-131; 'ghm' is not built), ``linkloss_type`` 'l2' or 'l1' (LinkLoss, cost.py:66-94;
    like the reference, the constructor's own default 'l2_softmax' fails cost.py:73's assert - pass the config's value).  Scores are device tensors of the training-mode forward (``forward_train``)."""

    def __init__(self, smooth_ratio=0, detloss_type='bce', endloss_type='l2', det_ratio=0.4, trans_ratio=0.4,
                 trans_last=False, linkloss_type='l2_softmax'):
        super().__init__()
        for name, v in (('detloss_type', detloss_type), ('endloss_type', endloss_type)):
            if not any(k in v for k in ('bce', 'l2', 'l1', 'ghm')):
                raise NotImplementedError("%s %r: 'bce', 'l2', 'l1' and 'ghm' are built" % (name, v))
        # 'ghm' (cost.py:105-110 -> modules/ghm_loss.py GHMC_Loss(bins=30, momentum=0.75)) keeps running per-bin counts
        # between calls: one state per DetLoss instance of the reference - `det_loss`, and `end_loss`, which serves the new
        # AND the end scores in that order (cost.py:154-155,164-166).  float64 [30] device tensors, created on first use.
        self.ghm_bins, self.ghm_momentum = 30, 0.75
        self._ghm_acc = {}
        # cost.py:73 - the reference's own default 'l2_softmax' trips this assert; its configs pass 'l2'
        assert linkloss_type in ['l1', 'l2']
        self.smooth_ratio, self.det_ratio, self.trans_ratio, self.trans_last = smooth_ratio, det_ratio, trans_ratio, trans_last
        self.detloss_type, self.endloss_type, self.linkloss_type = detloss_type, endloss_type, linkloss_type
        self.ops = None  # operator backend (HipOps unless a test injects another)

    def _ops(self):
        if self.ops is None:
            from .ops import HipOps
            self.ops = HipOps()
        return self.ops

    @staticmethod
    def _det_terms(si, score, gt, loss_type, ratio):
        """DetLoss.forward (cost.py:109-131): every type named in `loss_type` REPLACES the previous one (plain
        assignments in the reference), so the last matching one counts: order bce, l2, l1"""
        kind = None
        for k in ('bce', 'l2', 'l1', 'ghm'):
            if k in loss_type:
                kind = k
        R, C = score.shape
        gt = gt.to(torch.float32).contiguous()
        if kind == 'ghm':  # no reduction='mean' here: GHMC_Loss divides by its own count of valid elements
            return [(si, None, gt, 'ghm', ratio, dict(state='det' if si == 0 else 'end'))]
        mask = {} if kind == 'bce' else dict(mcol=gt, M=C, mask_mode=2, ignore=-1.0)
        return [(si, None, gt, LOSS_KINDS[kind], ratio / (R * C), mask)]

    def ghm_state(self, which, device=None):
        """running per-bin counts of the 'det' / 'end' GHMC_Loss (float64 [bins] device tensor; the reference's acc_sum)"""
        t = self._ghm_acc.get(which)
        if t is None:
            t = self._ghm_acc[which] = torch.zeros(self.ghm_bins, dtype=torch.float64, device=device)
        elif device is not None and t.device != torch.device(device):
            t = self._ghm_acc[which] = t.to(device)  # the momentum history moves with the criterion (it was re-zeroed)
        return t

    def forward(self, det_split, gt_det, gt_link, gt_new, gt_end, det_score, link_score, new_score, end_score, trans=None):
        split = [int(d.item()) if torch.is_tensor(d) else int(d) for d in det_split]
        if min(split) <= 0:
            # cost.py:166 slices gt_end[:-det_split[-1]] (EMPTY for a last frame without detections, where the slice
            # below would be the full vector) and a link term over N * M == 0 pairs has no mean: the reference's
            # training samples are built from frames that have detections (dataset/patchwise_dataset.py:205-216
            # keeps frames present in sequence_det), so such a sample is refused instead of guessed at
            raise ValueError('TrackingLoss: every frame of a sample needs at least one detection (det_split = %r)' % (split,))
        scores = [det_score, new_score, end_score] + list(link_score)
        terms = self._det_terms(0, det_score, gt_det, self.detloss_type, self.det_ratio)
        terms += self._det_terms(1, new_score, gt_new[split[0]:], self.endloss_type, 0.4)
        terms += self._det_terms(2, end_score, gt_end[:gt_end.shape[0] - split[-1]], self.endloss_type, 0.4)
        gtd = gt_det.to(torch.float32).contiguous()
        for i, lk in enumerate(link_score):
            R, N, M = lk.shape
            # NB cost.py:81-86: idx_base is never advanced, so every pair is masked with the FIRST frames' gt_det
            mask = dict(mrow=gtd[0:split[i]].contiguous(), mcol=gtd[split[i]:split[i] + split[i + 1]].contiguous(),
                        M=M, mask_mode=1)
            y = gt_link[i].to(torch.float32).reshape(-1).contiguous()
            for k in ('l2', 'l1'):
                if k in self.linkloss_type:
                    terms.append((3 + i, (R, N * M), y, LOSS_KINDS[k], 1.0 / (R * N * M), mask))
        for t in terms:  # 'ghm' terms: hand the kernel this module's state tensors
            if t[3] == 'ghm':
                t[5].update(state=self.ghm_state(t[5]['state'], det_score.device), bins=self.ghm_bins, momentum=self.ghm_momentum)
        loss = _ScoreLossFn.apply(self._ops(), terms, *scores)
        if trans is not None:
            # cost.py:175-184 (an ELEMENTWISE product with the transpose, as written there); 64 x 64: torch
            tl = range(len(trans)) if self.trans_last else [len(trans) - 1]
            for i in tl:
                eye = torch.eye(trans[i].size(-1), dtype=trans[i].dtype, device=trans[i].device)
                loss = loss + Fn.mse_loss(trans[i] * trans[i].transpose(-1, -2), eye.expand_as(trans[i])) * self.trans_ratio
        return loss


# ======================================================================================================================
# training-mode forward of the whole network (image branch frozen)
# ======================================================================================================================
def forward_train(model, dets, det_info, dets_split):
    """``TrackingNet.forward`` in training mode (reference modules/tracking_net.py:165-193 with ``self.training``):
    returns (det_scores 3 x L raw, [link_scores 3 x N x M ...], new_scores 3 x (L - N_first), end_scores 3 x (L - N_last),
    trans) on the autograd graph of every parameter (``model.freeze_appearance = True`` keeps the image encoder out)."""
    from . import torch_ops
    from .plan import BatchPlan
    fc = [int(d.item()) if torch.is_tensor(d) else int(d) for d in dets_split]
    ps = det_info['points_split'].reshape(-1).detach().to('cpu').numpy().astype(np.int64)
    points = det_info['points']
    points = points.reshape(-1, points.shape[-1]).contiguous()
    S = int(dets.shape[-1])
    # the plan (integer tile tables, uploaded once) and everything the backward caches on it (row tilings of the trunk's
    # layers, segment tables) are kept per sample layout - the key holds the whole points_split, because the tables depend
    # on it.  What this buys: loops that revisit a handful of samples (the overfit check, tools/bench_train.py, gradient
    # accumulation over a fixed mini-set) skip ~250 small host-to-device copies per step (rocprof, round 4).  What it
    # does not: a pass over a real dataset sees every layout once per epoch, so it never hits - hence a SMALL
    # least-recently-used cache (MMMOT_TRAIN_PLAN_CACHE, default 8 plans of < 2 MB of device tables each) instead of one
    # that pins 32 dead plans
    cache = model.__dict__.get('_train_plans')
    if cache is None:
        cache = model.__dict__['_train_plans'] = collections.OrderedDict()
    cap = max(int(os.environ.get('MMMOT_TRAIN_PLAN_CACHE', '8')), 1)
    key = (tuple(fc), ps.tobytes(), S, str(points.device))
    plan = cache.get(key)
    if plan is not None:
        cache.move_to_end(key)
    else:
        while len(cache) >= cap:
            cache.popitem(last=False)
        plan = cache[key] = BatchPlan([(fc, ps)], S, points.device, rows=(0, 1, 2), use_points=True)
    eng = _current_engine(model)
    if not getattr(model, 'freeze_appearance', False):
        from .train_vgg import appearance_autograd
        img = appearance_autograd(model, plan, dets)   # training-mode trunk: batch-statistics BatchNorm2d, differentiable
    else:                                              # frozen image branch: the eval-mode (folded) inference trunk
        with torch.no_grad():
            if eng.ops.name == 'hip':
                img = torch.ops.mmmot.appearance(dets.contiguous(), torch_ops.engine_handle(eng),
                                                 torch_ops.plan_handle(plan))
            else:  # an injected backend (tests: the torch emulation of the C-ABI)
                eng.dev = dets.device
                cat0 = eng.buf('cat', plan.Lt, 1024)
                eng.appearance(plan, dets.contiguous(), cat0)
                img = cat0[:, :512].clone()
    pts_feat, trans = pointnet_autograd(model, plan, points)
    cat = torch.cat([img, pts_feat], dim=1)
    det, link, new, end = head_autograd(model, plan, cat)
    nR = plan.nR
    links, pi = [], 0
    for f in range(len(fc) - 1):
        _, _, N, _, M = plan.pairs[pi]
        o = plan.link_off[pi]
        links.append(link[o:o + nR * N * M].view(nR, N, M))
        pi += 1
    L = sum(fc)
    return det, links, new[:, fc[0]:], end[:, :L - fc[-1]], trans
