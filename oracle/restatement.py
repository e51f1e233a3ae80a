"""ORACLE - test infrastructure, not product code.

CPU (PyTorch fp32) restatement of the reference's per-frame-pair network
forward, ``TrackingNet.forward`` in eval mode (reference
modules/tracking_net.py:165-193).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product
(``mmmot_amd``) never does.

Pinning: the reference ships no tests / golden vectors for this path (SURVEY
section 4), so the oracle is pinned against the REAL reference imported in the
build container: ``oracle/gen_golden.py`` loads the same generated weights into
``/root/reference/modules.TrackingNet`` and into this restatement, asserts
agreement (<= 2e-5 on every output) and writes the reference's outputs to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this file
against those fixtures wherever the repo runs.

It is written functionally over a reference-keyed ``state_dict`` - one
function per reference function, each citing the lines it follows.  Layout is
the reference's own (N,C,L / N,C,H,W), deliberately different from the HIP
path's position-major layout so that the two share no code.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def _gn(x, sd, key, groups):
    return F.group_norm(x, groups, sd[key + '.weight'], sd[key + '.bias'], EPS)


def _bn_buffers_after(x, sd, prefix, bn_stats, momentum=0.1):
    """What a training-mode nn.BatchNorm{1,2}d forward leaves in its buffers (torch.nn.modules.batchnorm, the module the
    reference builds at modules/vgg.py:75 / tracking_net.py:93,96 with the default momentum 0.1):
    running <- (1 - m) running + m * (batch mean | UNBIASED batch variance), num_batches_tracked + 1."""
    if bn_stats is None:
        return
    dims = [d for d in range(x.dim()) if d != 1]
    n = x.numel() // x.shape[1]
    with torch.no_grad():
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False) * (n / max(n - 1, 1))
        bn_stats[prefix + 'running_mean'] = (1 - momentum) * sd[prefix + 'running_mean'] + momentum * mean
        bn_stats[prefix + 'running_var'] = (1 - momentum) * sd[prefix + 'running_var'] + momentum * var
        bn_stats[prefix + 'num_batches_tracked'] = sd[prefix + 'num_batches_tracked'] + 1


def vgg_stage(x, sd, stage, training=False, bn_stats=None):
    """One regrouped VGG16-BN stage: reference modules/appear_net.py:130-157 (regrouping),
    modules/vgg.py:67-80 (conv3x3 pad1 -> BatchNorm2d -> ReLU, 'M' = MaxPool 2x2).  ``training``: BatchNorm2d on the
    statistics of the batch (the module under .train(), tracking_model.py:41-48) instead of the running buffers;
    ``bn_stats`` (a dict) then receives the buffers the layer holds AFTER the forward."""
    layout = {0: [64, 64, 'M', 128, 128, 'M'], 1: [256, 256, 256, 'M'], 2: [512, 512, 512, 'M'],
              3: [512, 512, 512, 'M']}[stage]
    p = 'appearance.layers.%d.' % stage
    idx = 0
    for v in layout:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.conv2d(x, sd[p + '%d.weight' % idx], sd[p + '%d.bias' % idx], padding=1)
            b = p + '%d.' % (idx + 1)
            if training:
                _bn_buffers_after(x, sd, b, bn_stats)
                x = F.batch_norm(x, None, None, sd[b + 'weight'], sd[b + 'bias'], True, 0.0, EPS)
            else:
                x = F.batch_norm(x, sd[b + 'running_mean'], sd[b + 'running_var'], sd[b + 'weight'], sd[b + 'bias'],
                                 False, 0.0, EPS)
            x = F.relu(x)
            idx += 3
    return x


DROPBLOCK_PROB = 0.1  # reference modules/dropblock.py:22 (DropBlock2D's default; appear_net.py:18 passes the size only)


def dropblock2d(x, block_size, drop_prob=DROPBLOCK_PROB):
    """reference modules/dropblock.py:28-68 (DropBlock2D.forward in training mode): a Bernoulli(gamma) seed mask per
    (sample, pixel) - drawn on the HOST with the global torch generator, like the reference (`torch.rand(...)` then
    `.to(x.device)`) -, grown to block_size x block_size blocks by a max pool, the kept pixels rescaled by numel / sum."""
    gamma = drop_prob / (block_size ** 2)
    mask = (torch.rand(x.shape[0], *x.shape[2:]) < gamma).float().to(x.device)
    bm = F.max_pool2d(mask[:, None, :, :], kernel_size=(block_size, block_size), stride=(1, 1), padding=block_size // 2)
    if block_size % 2 == 0:
        bm = bm[:, :, :-1, :-1]
    bm = 1 - bm.squeeze(1)
    out = x * bm[:, None, :, :]
    return out * bm.numel() / bm.sum()


def skippool(x, sd, stage, dropblock=0):
    """reference modules/appear_net.py:9-32: (DropBlock, training mode, stages whose SkipPool got a block size) ->
    global avg pool -> GN(1) -> 1x1 -> GN(1) -> ReLU -> 1x1 -> GN(1) -> ReLU."""
    p = 'appearance.global_pool.%d.fc.' % stage
    if dropblock:
        x = dropblock2d(x, dropblock)
    o = F.adaptive_avg_pool2d(x, 1)
    o = _gn(o, sd, p + '0', 1)
    o = F.conv2d(o, sd[p + '1.weight'], sd[p + '1.bias'])
    o = F.relu(_gn(o, sd, p + '2', 1))
    o = F.conv2d(o, sd[p + '4.weight'], sd[p + '4.bias'])
    o = F.relu(_gn(o, sd, p + '5', 1))
    return o.flatten(1)


def appearance(crops, sd, keep=None, training=False, bn_stats=None, dropblock=0):
    """reference modules/appear_net.py:166-190 (vgg + skippool path): L x 3 x S x S -> L x 512.  ``dropblock`` (the
    block size, training mode only): `_parse_vgg_layers` (appear_net.py:130-157) hands it to the SkipPools made after the
    fourth and fifth max pool - stages 2 and 3; the first two stages get block size 0 = no DropBlock."""
    outs = []
    x = crops
    for s in range(4):
        x = vgg_stage(x, sd, s, training, bn_stats)
        if keep is not None:
            keep['vgg_stage%d' % s] = x
        outs.append(skippool(x, sd, s, dropblock if (training and s >= 2) else 0))
    return torch.cat(outs, dim=-1)


def stn_transform(sd, prefix, k):
    """reference modules/point_net.py:72-86.  fc_bn1 / fc_bn2 are GroupNorm(C, C) applied to a 1 x C
    tensor: one value per group, so the normalised value is exactly 0 and the layer returns its
    bias; the conv trunk and the global max therefore never reach the output:
        trans = output(relu(fc_bn2.bias)) + I
    (verified against the imported reference in oracle/gen_golden.py)."""
    h = F.relu(sd[prefix + 'fc_bn2.bias'])
    t = F.linear(h, sd[prefix + 'output.weight'], sd[prefix + 'output.bias'])
    return t.view(1, k, k) + sd[prefix + 'idt']


def _segment_avg(x, split):
    """per-detection average over point ranges: reference modules/point_net.py:32-39 / 139-146."""
    cols = [x[:, :, int(split[i]):int(split[i + 1])].mean(dim=-1, keepdim=True) for i in range(len(split) - 1)]
    return torch.cat(cols, dim=-1)


def pointnet(points, split, sd, keep=None):
    """reference modules/point_net.py:25-44 + 115-153.  points 1 x 3 x P, split (L+1,) -> L x 512, trans."""
    q = 'point_net.feat.'
    t1 = stn_transform(sd, q + 'stn1.', points.shape[1])  # 3 x 3, or 4 x 4 with the reflectivity channel
    x = torch.bmm(points.transpose(2, 1), t1).transpose(2, 1)                       # :119-123
    x = F.relu(_gn(F.conv1d(x, sd[q + 'conv1.weight'], sd[q + 'conv1.bias']), sd, q + 'bn1', 64))  # :125
    t2 = stn_transform(sd, q + 'stn2.', 64)
    x = torch.bmm(x.transpose(2, 1), t2).transpose(2, 1)                            # :127-131
    skip = x                                                                         # :132
    for i, g in ((2, 64), (3, 64), (4, 128), (5, 1024)):                             # :134-138
        x = F.relu(_gn(F.conv1d(x, sd[q + 'conv%d.weight' % i], sd[q + 'conv%d.bias' % i]), sd, q + 'bn%d' % i, g))
    avg = _segment_avg(x, split)                                                     # 1 x 1024 x L  (:139-146)
    if keep is not None:
        keep['pn_seg1024'] = avg[0].t()
    counts = (split[1:] - split[:-1]).long()
    rep = torch.repeat_interleave(avg, counts, dim=-1)                               # broadcast back to points (:145)
    x = torch.cat([skip, rep], dim=1)                                                # 1088 ch (point_net.py:26-27)
    x = F.relu(_gn(F.conv1d(x, sd['point_net.conv1.weight'], sd['point_net.conv1.bias']), sd, 'point_net.bn1', 512))
    x = _segment_avg(x, split)                                                       # :32-39
    x = F.relu(_gn(F.conv1d(x, sd['point_net.conv2.weight'], sd['point_net.conv2.bias']), sd, 'point_net.bn2', 16))
    return x[0].t(), [t1, t2]


def fusion(feats, sd, mode):
    """reference modules/fusion_net.py:31-42 (C), 62-70 (B), 85-92 (A).  feats 1 x 1024 x L -> 3 x 512 x L."""
    p = 'fusion_module.'
    two = feats.view(2, -1, feats.size(-1))
    img, pts = two[:1], two[1:]  # feats[0] = image features; the *_p parameters act on them (naming trap)
    lin = lambda x, name: _gn(F.conv1d(x, sd[p + name + '.0.weight'], sd[p + name + '.0.bias']), sd, p + name + '.1', 512)
    if mode == 'A':
        fused = lin(feats, 'input_w')
    elif mode == 'B':
        fused = lin(img, 'input_p') + lin(pts, 'input_i')
    else:
        gp = torch.sigmoid(F.conv1d(img, sd[p + 'gate_p.0.weight'], sd[p + 'gate_p.0.bias']))
        gi = torch.sigmoid(F.conv1d(pts, sd[p + 'gate_i.0.weight'], sd[p + 'gate_i.0.bias']))
        fused = (gp * lin(img, 'input_p') + gi * lin(pts, 'input_i')) / (gp + gi)
    return torch.cat([two, fused], dim=0)


def det_head(feats, sd, score_arch, neg_threshold):
    """reference modules/tracking_net.py:91-100 (w_det) + 149-163 (eval branch)."""
    x = feats
    for i, bn in ((0, 1), (3, 4)):
        x = F.conv1d(x, sd['w_det.%d.weight' % i], sd['w_det.%d.bias' % i])
        b = 'w_det.%d.' % bn
        x = F.relu(F.batch_norm(x, sd[b + 'running_mean'], sd[b + 'running_var'], sd[b + 'weight'], sd[b + 'bias'],
                                False, 0.0, EPS))
    s = F.conv1d(x, sd['w_det.6.weight'], sd['w_det.6.bias']).squeeze(1)
    if 'cls' in score_arch:
        s = torch.sigmoid(s)
    return s - (s < neg_threshold).float()


def pairwise(a, b, op):
    """reference modules/gcn.py:6-14 / 17-28 / 31-41: R x C x N , R x C x M -> R x C x N x M."""
    if op == 'multiply':
        return a.unsqueeze(-1) * b.unsqueeze(-2)
    d = (a.unsqueeze(-1) - b.unsqueeze(-2)) / 2
    return d.abs() if op == 'minus_abs' else d


def new_end(x, sd, keep=None, mode='avg'):
    """reference modules/new_end.py:62-82 (v2): mode 'avg' means over the prev / curr axis (:69-71), anything else
    takes the maximum (:72-74)."""
    p = 'w_link.w_new_end.'
    x = F.relu(_gn(F.conv2d(x, sd[p + 'conv0.0.weight'], sd[p + 'conv0.0.bias']), sd, p + 'conv0.1', 1))
    if mode == 'avg':
        new_vec, end_vec = x.mean(dim=-2), x.mean(dim=-1)
    else:
        new_vec, end_vec = x.max(dim=-2)[0], x.max(dim=-1)[0]

    def head(v):
        v = F.relu(_gn(F.conv1d(v, sd[p + 'conv1.0.weight'], sd[p + 'conv1.0.bias']), sd, p + 'conv1.1', 1))
        v = F.relu(_gn(F.conv1d(v, sd[p + 'conv1.3.weight'], sd[p + 'conv1.3.bias']), sd, p + 'conv1.4', 1))
        return torch.sigmoid(F.conv1d(v, sd[p + 'conv1.6.weight'], sd[p + 'conv1.6.bias'])).squeeze(1)

    if keep is not None:
        keep['new_vec'], keep['end_vec'] = new_vec, end_vec
    return head(new_vec), head(end_vec)


def affinity(a, b, sd, op, keep=None, end_mode='avg'):
    """reference modules/gcn.py:68-82: link logits R x 1 x N x M, new R x M, end R x N."""
    x = pairwise(a, b, op)
    new, end = new_end(x, sd, keep, end_mode)
    p = 'w_link.conv1.'
    for i, g in ((0, 512), (3, 512), (6, 128)):
        x = F.relu(_gn(F.conv2d(x, sd[p + '%d.weight' % i], sd[p + '%d.bias' % i]), sd, p + '%d' % (i + 1), g))
    x = F.conv2d(x, sd[p + '9.weight'], sd[p + '9.bias'])
    return x, new, end


def softmax_mode(link, mode):
    """reference modules/tracking_net.py:106-126."""
    if mode == 'single':
        return F.softmax(link, dim=-1)
    if mode in ('dual', 'dual_add', 'dual_max'):
        p, q = F.softmax(link, dim=-1), F.softmax(link, dim=-2)
        return p * q if mode == 'dual' else (p + q) / 2 if mode == 'dual_add' else torch.max(p, q)
    return link


def tracking_forward(sd, cfg, dets, points, points_split, dets_split, keep=None, rows=(0, 1, 2)):
    """Eval-mode ``TrackingNet.forward`` (reference modules/tracking_net.py:128-193).

    sd: reference-keyed state_dict (fp32 CPU); cfg: dict(fusion, affinity_op, softmax_mode,
    neg_threshold, score_arch); dets L x 3 x S x S; points 1 x P x 3; points_split (L+1,);
    dets_split list of frame counts.  rows != (0,1,2) evaluates single-modality rows only
    (rows are independent: SURVEY 8a), skipping the unused encoder."""
    rows = tuple(rows)
    trans = None
    feats = {}
    if 0 in rows or 2 in rows:
        feats[0] = appearance(dets, sd, keep)                                   # :131-133
    if 1 in rows or 2 in rows:
        split = points_split.reshape(-1).long()
        feats[1], trans = pointnet(points.transpose(-1, -2), split, sd, keep)    # :136-140
    if rows == (0, 1, 2):
        cat = torch.cat([feats[0], feats[1]], dim=-1).t().unsqueeze(0)           # :142
        F3 = fusion(cat, sd, cfg['fusion'])                                      # :143-145
    else:
        F3 = torch.stack([feats[r].t() for r in rows], dim=0)
    if keep is not None:
        keep['F'] = F3
    det = det_head(F3, sd, cfg.get('score_arch', 'branch_cls'), cfg['neg_threshold'])  # :167
    counts = [int(c) for c in dets_split]
    links, news, ends = [], [], []
    start = 0
    for i in range(len(counts) - 1):                                             # :173-181
        mid, stop = start + counts[i], start + counts[i] + counts[i + 1]
        logit, new, end = affinity(F3[:, :, start:mid], F3[:, :, mid:stop], sd, cfg['affinity_op'], keep,
                                   cfg.get('end_mode', 'avg'))
        links.append(softmax_mode(logit, cfg['softmax_mode']).squeeze(1))
        news.append(new)
        ends.append(end)
        start = mid
    R = F3.size(0)
    new = torch.cat([F3.new_zeros(R, counts[0])] + news, dim=1)                  # :183-189 (eval padding)
    end = torch.cat(ends + [F3.new_zeros(R, counts[-1])], dim=1)
    return det, links, new, end, trans


# ======================================================================================================================
# Training step (reference tracking_model.py:50-66): training-mode forward + TrackingLoss.  Checker for
# mmmot_amd/train.py / backward.py: torch.autograd through THESE functions (in float64) is the gradient reference.
# ======================================================================================================================
def det_head_train(feats, sd, bn_stats=None):
    """reference modules/tracking_net.py:91-100 + 149-151 with ``self.training``: BatchNorm1d on the statistics of the
    batch (the 3 modality rows x L positions of the one sample), raw scores (no sigmoid, no threshold mask)."""
    x = feats
    for i, bn in ((0, 1), (3, 4)):
        x = F.conv1d(x, sd['w_det.%d.weight' % i], sd['w_det.%d.bias' % i])
        _bn_buffers_after(x, sd, 'w_det.%d.' % bn, bn_stats)
        x = F.relu(F.batch_norm(x, None, None, sd['w_det.%d.weight' % bn], sd['w_det.%d.bias' % bn], True, 0.0, EPS))
    return F.conv1d(x, sd['w_det.6.weight'], sd['w_det.6.bias']).squeeze(1)


def tracking_forward_train(sd, cfg, img_feats, points, points_split, dets_split, crops=None, bn_stats=None, dropblock=0):
    """Training-mode ``TrackingNet.forward`` (reference modules/tracking_net.py:165-193 with ``self.training``): image
    encoder in training mode on ``crops`` (batch-statistics BatchNorm2d), or - ``crops`` None - GIVEN image features
    ``img_feats`` L x 512 (the product's frozen-image-branch mode); PointNet (GroupNorm only: identical in both modes),
    fusion, training-mode w_det, the pairwise block; new / end scores are NOT padded in training mode (:190-192).
    ``bn_stats`` (a dict): receives the BatchNorm buffers (trunk and w_det) as the modules hold them after this forward.
    Pinned to the imported reference's training step by oracle/gen_golden_train.py (tests/golden/train_*.npz)."""
    if crops is not None:
        img_feats = appearance(crops, sd, training=True, bn_stats=bn_stats, dropblock=dropblock)
    split = points_split.reshape(-1).long()
    pts, trans = pointnet(points.transpose(-1, -2), split, sd)
    cat = torch.cat([img_feats, pts], dim=-1).t().unsqueeze(0)
    F3 = fusion(cat, sd, cfg['fusion'])
    det = det_head_train(F3, sd, bn_stats)
    counts = [int(c) for c in dets_split]
    links, news, ends = [], [], []
    start = 0
    for i in range(len(counts) - 1):
        mid, stop = start + counts[i], start + counts[i] + counts[i + 1]
        logit, new, end = affinity(F3[:, :, start:mid], F3[:, :, mid:stop], sd, cfg['affinity_op'], None,
                                   cfg.get('end_mode', 'avg'))
        links.append(softmax_mode(logit, cfg['softmax_mode']).squeeze(1))
        news.append(new)
        ends.append(end)
        start = mid
    return det, links, torch.cat(news, dim=1), torch.cat(ends, dim=1), trans


GHM_BINS, GHM_MOMENTUM = 30, 0.75  # reference cost.py:108: GHMC_Loss(bins=30, momentum=0.75)


def new_ghm_state(bins=GHM_BINS):
    """the running per-bin counts a GHMC_Loss keeps between calls (reference modules/ghm_loss.py:25-26: acc_sum)"""
    return [0.0] * bins


def ghmc_loss(score, target, mask, state, bins=GHM_BINS, momentum=GHM_MOMENTUM):
    """reference modules/ghm_loss.py:28-61 (GHMC_Loss.forward): gradient-harmonised binary cross entropy.  g = |sigmoid(x) -
    y| is histogrammed over `bins` equal bins of [0, 1] (last edge + 1e-6), every valid element is weighted with
    tot / (running count of its bin) / (number of non-empty bins), the weights are constants of the graph.  `state` (the
    running counts, a list of `bins` floats) is UPDATED in place: count <- momentum * count + (1 - momentum) * n_in_bin."""
    edges = [float(x) / bins for x in range(bins + 1)]
    edges[-1] += 1e-6
    weights = torch.zeros_like(score)
    g = torch.abs(score.sigmoid().detach() - target)
    valid = mask > 0
    tot = max(valid.float().sum().item(), 1.0)
    n = 0
    for i in range(bins):
        inds = (g >= edges[i]) & (g < edges[i + 1]) & valid
        num_in_bin = inds.sum().item()
        if num_in_bin > 0:
            if momentum > 0:
                state[i] = momentum * state[i] + (1 - momentum) * num_in_bin
                weights[inds] = tot / state[i]
            else:
                weights[inds] = tot / num_in_bin
            n += 1
    if n > 0:
        weights = weights / n
    return F.binary_cross_entropy_with_logits(score, target, weights, reduction='sum') / tot


def _det_loss(score, gt, loss_type, ignore_index=-1, ghm_state=None):
    """reference cost.py:109-131 (DetLoss.forward): later types overwrite earlier ones.  'ghm': `ghm_state` is the
    DetLoss instance's GHMC_Loss state (new_ghm_state(); the caller keeps it between calls like the module does)"""
    gt = gt.unsqueeze(0).repeat(score.size(0), 1)
    loss = None
    if 'bce' in loss_type:
        loss = F.binary_cross_entropy_with_logits(score, gt)
    if 'l2' in loss_type:
        mask = 1 - gt.eq(ignore_index).to(score.dtype)
        loss = F.mse_loss(score.mul(mask), gt)
    if 'l1' in loss_type:
        mask = 1 - gt.eq(ignore_index).to(score.dtype)
        loss = F.smooth_l1_loss(score.mul(mask), gt)
    if 'ghm' in loss_type:
        mask = 1 - gt.eq(ignore_index).to(score.dtype)
        loss = ghmc_loss(score, gt, mask, ghm_state)
    return loss


def _link_loss(det_split, gt_det, link_score, gt_link, loss_type):
    """reference cost.py:77-94 (LinkLoss.forward).  idx_base is never advanced there: every pair is masked with the
    first frames' gt_det - restated as written."""
    loss = 0
    idx_base = 0
    for i in range(len(link_score)):
        curr_num, next_num = int(det_split[i]), int(det_split[i + 1])
        mask = torch.ones_like(link_score[i])
        curr = (gt_det[idx_base:idx_base + curr_num] == 1).to(mask.dtype)
        nxt = (gt_det[idx_base + curr_num:idx_base + curr_num + next_num] == 1).to(mask.dtype)
        mask = mask * curr.unsqueeze(-1) * nxt.unsqueeze(0)
        tgt = gt_link[i].reshape(1, curr_num, next_num).repeat(mask.size(0), 1, 1)
        if 'l2' in loss_type:
            loss = loss + F.mse_loss(link_score[i].mul(mask), tgt)
        if 'l1' in loss_type:
            loss = loss + F.smooth_l1_loss(link_score[i].mul(mask), tgt)
    return loss


def tracking_loss(det_split, gt_det, gt_link, gt_new, gt_end, det_score, link_score, new_score, end_score, trans=None,
                  detloss_type='bce', endloss_type='l2', det_ratio=0.4, trans_ratio=0.4, trans_last=False,
                  linkloss_type='l2_softmax', ghm_state=None):
    """reference cost.py:160-185 (TrackingLoss.forward).  ``ghm_state``: dict(det=..., end=...) of new_ghm_state() lists
    for 'ghm' loss types - the state of the module's two DetLoss instances (cost.py:154-155; the end instance serves the
    new AND the end scores, in that order), carried from call to call by the caller."""
    split = [int(d) for d in det_split]
    gs = ghm_state or {}
    loss = _det_loss(det_score, gt_det, detloss_type, ghm_state=gs.get('det')) * det_ratio
    loss = loss + _det_loss(new_score, gt_new[split[0]:], endloss_type, ghm_state=gs.get('end')) * 0.4
    loss = loss + _det_loss(end_score, gt_end[:gt_end.shape[0] - split[-1]], endloss_type, ghm_state=gs.get('end')) * 0.4
    loss = loss + _link_loss(split, gt_det, link_score, gt_link, linkloss_type)
    if trans is not None:
        idx = range(len(trans)) if trans_last else [len(trans) - 1]
        for i in idx:
            eye = torch.eye(trans[i].size(-1), dtype=trans[i].dtype)
            loss = loss + F.mse_loss(trans[i] * trans[i].transpose(-1, -2), eye.expand_as(trans[i])) * trans_ratio
    return loss
