"""Training backward of the pairwise block (first slice).

Gradients of ``affinity_module.forward`` + ``NewEndIndicator_v2.forward`` + the softmax modes of
``TrackingNet.associate`` (reference modules/gcn.py:68-82, new_end.py:62-82, tracking_net.py:106-126) with respect
to the fused features F (3 x L x 512 in the reference, [nR*Lt][512] here) and to every ``w_link.*`` parameter - the
part of the training step ``tracking_model.py:50-66`` (forward -> loss -> backward) that runs over the N x M pair
space.  In training mode this block is identical to eval mode (GroupNorm only: no BatchNorm, no dropout), so the
forward is the inference schedule of ``Engine.affinity`` with the pre-norm tensors kept on a tape.

    link, new, end = affinity_autograd(model, plan, F)        # F requires grad / w_link parameters require grad
    loss(link, new, end).backward()                           # fills F.grad and model.w_link.*.grad

Everything numeric happens in libmmmot_hip.so (csrc/backward.hip + the forward kernels); torch does memory,
views, transposes of weights (data movement) and the autograd bookkeeping.  Not built yet (next slices): the
backward of w_det, fusion, PointNet and the VGG trunk (training-mode BatchNorm), the losses of cost.py:134-185.
"""
import collections

import numpy as np
import torch

from .engine import EPS
from .ops import ACT_NONE, ACT_SIGMOID, A_NORM_RELU, A_PAIR, A_PLAIN, PAIR_OPS, SOFTMAX_MODES
from .plan import Segments

# packed name -> reference state_dict key (w_link.*); 'wa' / 'ba' are the stacked [new_end.conv0 ; conv1.0] layer
PARAM_KEYS = {
    'g_ne0': 'w_link.w_new_end.conv0.1.weight', 'be_ne0': 'w_link.w_new_end.conv0.1.bias',
    'g1': 'w_link.conv1.1.weight', 'be1': 'w_link.conv1.1.bias',
    'w3': 'w_link.conv1.3.weight', 'b3': 'w_link.conv1.3.bias',
    'g4': 'w_link.conv1.4.weight', 'be4': 'w_link.conv1.4.bias',
    'w6': 'w_link.conv1.6.weight', 'b6': 'w_link.conv1.6.bias',
    'g7': 'w_link.conv1.7.weight', 'be7': 'w_link.conv1.7.bias',
    'w9': 'w_link.conv1.9.weight', 'b9': 'w_link.conv1.9.bias',
    'nw0': 'w_link.w_new_end.conv1.0.weight', 'nb0': 'w_link.w_new_end.conv1.0.bias',
    'ng1': 'w_link.w_new_end.conv1.1.weight', 'nbe1': 'w_link.w_new_end.conv1.1.bias',
    'nw3': 'w_link.w_new_end.conv1.3.weight', 'nb3': 'w_link.w_new_end.conv1.3.bias',
    'ng4': 'w_link.w_new_end.conv1.4.weight', 'nbe4': 'w_link.w_new_end.conv1.4.bias',
    'nw6': 'w_link.w_new_end.conv1.6.weight', 'nb6': 'w_link.w_new_end.conv1.6.bias',
}


class _PlanAux:
    """Integer tables of the backward that the forward plan does not carry (built once per plan)."""

    def __init__(self, plan):
        dev = plan.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        gN, gM = plan.h_pg_N, plan.h_pg_M
        # V rows of group g: M "new" rows then N "end" rows (plan.v_tiles has two groups per pair group)
        self.vrow0 = up(np.concatenate([[0], np.cumsum(gN + gM)])[:-1])
        a_g, a_i, b_g, b_j = [], [], [], []
        for g, (n, m) in enumerate(zip(gN, gM)):
            a_g += [g] * int(n); a_i += list(range(int(n)))
            b_g += [g] * int(m); b_j += list(range(int(m)))
        self.a_grp, self.a_idx, self.b_grp, self.b_idx = up(a_g), up(a_i), up(b_g), up(b_j)
        self._tsum = {}

    def tile_sums(self, tiles, device):
        """(per-group, total) segment tables that sum the [T][2][C] partials of a tiling, viewed as [2T][C] rows.  Each is
        a Segments, or - long segments - a two-level (chunks, chunk sums) pair for _segsum: one segment is one workgroup,
        and the 8 624 tiles of a full-resolution trunk layer walked by two workgroups were the 0.9 ms launches of the
        training step's kernel table (profiles/r05/rocprofv3_kernel_stats_train.txt)."""
        key = id(tiles)
        if key not in self._tsum:
            G = tiles.G
            start = np.stack([2 * tiles.h_g_tile0, 2 * tiles.h_g_tile0 + 1], 1).reshape(-1)
            count = np.repeat(tiles.h_g_ntiles, 2)
            grp = _chunked_segments(start, count, 2, device)
            tot = _chunked_segments(np.array([0, 1]), np.array([tiles.T, tiles.T]), 2, device)
            self._tsum[key] = (grp, tot)
        return self._tsum[key]


SEGSUM_CHUNK = 128  # rows per first-level segment of a long strided sum


def _chunked_segments(start, count, stride, device):
    """sum (divisor 1) of `count[i]` rows from `start[i]` with `stride`: one Segments, or - any segment longer than
    4 * SEGSUM_CHUNK rows - (first level: SEGSUM_CHUNK-row chunks of every segment, second level: a segment's chunk sums)"""
    start, count = np.asarray(start, np.int64), np.asarray(count, np.int64)
    n = len(start)
    if n == 0 or count.max() <= 4 * SEGSUM_CHUNK:
        return Segments(start, count, np.full(n, stride), np.zeros(n), device, div=np.ones(n))
    nch = -(-count // SEGSUM_CHUNK)
    s1, c1 = [], []
    for i in range(n):
        k = np.arange(nch[i])
        s1.append(start[i] + stride * SEGSUM_CHUNK * k)
        c1.append(np.minimum(SEGSUM_CHUNK, count[i] - SEGSUM_CHUNK * k))
    s1, c1 = np.concatenate(s1), np.concatenate(c1)
    first = Segments(s1, c1, np.full(len(s1), stride), np.zeros(len(s1)), device, div=np.ones(len(s1)))
    off = np.concatenate([[0], np.cumsum(nch)[:-1]])
    second = Segments(off, nch, np.ones(n), np.zeros(n), device, div=np.ones(n))
    return (first, second)


def _segsum(eng, X, C, segs, out):
    """segment sums through mmmot_segment_mean (divisor 1); `segs`: a Segments or a two-level pair of _chunked_segments"""
    if isinstance(segs, tuple):
        part = torch.empty(segs[0].n, C, dtype=torch.float32, device=X.device)
        eng.ops.segment_mean(X, C, segs[0], part, use_group=False)
        eng.ops.segment_mean(part, C, segs[1], out, use_group=False)
    else:
        eng.ops.segment_mean(X, C, segs, out, use_group=False)


def _aux(plan):
    if not hasattr(plan, '_bwd_aux'):
        plan._bwd_aux = _PlanAux(plan)
    return plan._bwd_aux


def _unit(eng, C, dev):
    key = ('unit', C, str(dev))
    if key not in eng.ws:
        eng.ws[key] = (torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev))
    return eng.ws[key]


def dgrad_gemm(eng, W, tiles, X, Y):
    """Y = X W for a forward weight W [N_f][K_f] (the input gradient dA_in = dY W of a 1x1 layer): the row GEMM with the
    transposed weight, exact fp32 matrix cores.  (Not the f16x3 row GEMM: its A operand - here the GRADIENT dY, 1e-4 ..
    1e-7 - is split into fp16 halves unscaled, i.e. into fp16's subnormals; measured: the SGD-step parity went from 2e-6
    to > 1e-5.  The weight-gradient GEMM scales dY by its maximum first, mmmot_gemm_tn_f16.)  The transposed copy is cached
    on the engine per (storage, version) of W, so a backward does not re-transpose an unchanged weight.  An entry keeps
    a reference to W itself: the key contains W's ADDRESS, and a weight that is freed (the folded PointNet weights are
    fresh tensors every step) would hand its address - at version 0 - to the next step's fold, which would then hit the
    previous step's transpose (tests/test_train_cpu.py runs three steps against the oracle)."""
    Nf, Kf = int(W.shape[0]), int(W.shape[1])
    cache = eng.__dict__.get('_wt_cache')
    if cache is None:
        cache = eng.__dict__['_wt_cache'] = collections.OrderedDict()
    key = (W.data_ptr(), W._version, Nf, Kf)
    ent = cache.get(key)
    if ent is not None:
        cache.move_to_end(key)  # least-recently-USED eviction: a persistent weight that is hit every step stays
    else:
        # ~30 weights pass through per step (the folded PointNet weights are fresh tensors every step, the head weights
        # change version with every optimizer step): room for two steps, evict the entry unused for longest
        while len(cache) >= 64:
            cache.popitem(last=False)
        ent = cache[key] = (W, W.detach().t().contiguous())  # data movement; W pinned while the entry lives
    wt = ent[1]
    eng.ops.gemm(wt, tiles, Kf, Nf, X=X, Y=Y)


class _Layer:
    """One 'GEMM -> GroupNorm -> ReLU' layer on the tape: pre-norm output Y, its statistics, its parameters."""

    def __init__(self, Y, C, NG, gamma, beta, sc, sh, sc1, sh1, tiles):
        self.Y, self.C, self.NG, self.gamma, self.beta = Y, C, NG, gamma, beta
        self.sc, self.sh, self.sc1, self.sh1, self.tiles = sc, sh, sc1, sh1, tiles


def _norm_layer(eng, part, tiles, Y, C, NG, gamma, beta):
    """finalize twice: (gamma, beta) -> sc/sh for the consumer's prologue, (1, 0) -> sc1 = rstd, sh1 = -mean*rstd"""
    dev = Y.device
    new = lambda: torch.empty(tiles.G, C, dtype=torch.float32, device=dev)
    sc, sh, sc1, sh1 = new(), new(), new(), new()
    one, zero = _unit(eng, C, dev)
    eng.ops.gn_finalize(part, tiles, C, NG, gamma, beta, EPS, sc, sh)
    eng.ops.gn_finalize(part, tiles, C, NG, one, zero, EPS, sc1, sh1)
    return _Layer(Y, C, NG, gamma, beta, sc, sh, sc1, sh1, tiles)


def affinity_forward_train(eng, plan, F):
    """Engine.affinity with the tape the backward needs.  F: [nR, Lt, 512] (contiguous).  Returns
    (link flat [R], new [nR, Lt], end [nR, Lt], tape)."""
    ops, lk, PT, VT = eng.ops, eng.P['w_link'], plan.pair_tiles, plan.v_tiles
    if eng.end_mode != 'avg':
        raise NotImplementedError("the backward of end_mode='max' is not built")
    nR, Lt, R = plan.nR, plan.Lt, plan.pair_tiles.R
    dev = F.device
    Ff = F.reshape(nR * Lt, 512)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    pair = dict(row0=PT.g_row0, M=plan.pg_M, aoff=plan.pg_aoff, boff=plan.pg_boff)
    pairop = PAIR_OPS[eng.affinity_op]
    t = dict(pair=pair, pairop=pairop)
    # stacked [new_end.conv0 ; conv1.0] over the on-the-fly pairwise tensor
    ya, part = new(R, 1024), new(PT.T, 2, 1024)
    eng._gemm(lk, 'wa', PT, 1024, 512, FA=Ff, FB=Ff, pair=pair, amode=A_PAIR, pairop=pairop, bias=lk['ba'], Y=ya,
              part=part)
    t['ne0'] = _norm_layer(eng, part[:, :, 0:512], PT, ya[:, 0:512], 512, 1, lk['g_ne0'], lk['be_ne0'])
    t['l1'] = _norm_layer(eng, part[:, :, 512:1024], PT, ya[:, 512:1024], 512, 512, lk['g1'], lk['be1'])
    # new / end vectors and their head
    V = new(VT.R, 512)
    ops.segment_mean(ya[:, 0:512], 512, plan.v_segs, V, sc=t['ne0'].sc, sh=t['ne0'].sh, relu=True)
    vh0, part = new(VT.R, 512), new(VT.T, 2, 512)
    eng._gemm(lk, 'nw0', VT, 512, 512, X=V, bias=lk['nb0'], Y=vh0, part=part)
    t['v1'] = _norm_layer(eng, part, VT, vh0, 512, 1, lk['ng1'], lk['nbe1'])
    vh1, part = new(VT.R, 128), new(VT.T, 2, 128)
    eng._gemm(lk, 'nw3', VT, 128, 512, X=vh0, bias=lk['nb3'], Y=vh1, part=part, sc=t['v1'].sc, sh=t['v1'].sh,
              amode=A_NORM_RELU)
    t['v4'] = _norm_layer(eng, part, VT, vh1, 128, 1, lk['ng4'], lk['nbe4'])
    ne = torch.zeros(2, nR, Lt, dtype=torch.float32, device=dev)
    ops.rowdot(vh1, 128, lk['nw6'], lk['nb6'], VT, ne.view(-1), sc=t['v4'].sc, sh=t['v4'].sh, act=ACT_SIGMOID,
               omap=plan.v_omap)
    # link branch
    y3, part = new(R, 512), new(PT.T, 2, 512)
    eng._gemm(lk, 'w3', PT, 512, 512, X=ya[:, 512:1024], bias=lk['b3'], Y=y3, part=part, sc=t['l1'].sc, sh=t['l1'].sh,
              amode=A_NORM_RELU)
    t['l4'] = _norm_layer(eng, part, PT, y3, 512, 512, lk['g4'], lk['be4'])
    y6, part = new(R, 128), new(PT.T, 2, 128)
    eng._gemm(lk, 'w6', PT, 128, 512, X=y3, bias=lk['b6'], Y=y6, part=part, sc=t['l4'].sc, sh=t['l4'].sh,
              amode=A_NORM_RELU)
    t['l7'] = _norm_layer(eng, part, PT, y6, 128, 128, lk['g7'], lk['be7'])
    logits = new(R)
    ops.rowdot(y6, 128, lk['w9'], lk['b9'], PT, logits, sc=t['l7'].sc, sh=t['l7'].sh)
    link = logits
    if eng.softmax_mode != 'none':
        link = torch.empty_like(logits)
        ops.softmax_pairs(logits, link, PT.g_row0, plan.pg_N, plan.pg_M, PT.G, plan.max_nm,
                          SOFTMAX_MODES[eng.softmax_mode])
    t.update(ya=ya, V=V, logits=logits)
    return link, ne[0], ne[1], t


COLSUM_CHUNK = 128  # rows per segment of one level of _colsum


def _colsum(eng, X):
    """column sums of a [rows][C] tensor (C % 4 == 0) through the strided-mean kernel with divisor 1.  One segment = one
    workgroup per 256 channels, and a workgroup walks its rows sixteen at a time: the 2 048-row chunks this used to cut
    took 180 us each however few there were (the bias gradients of the trunk: 2.6 ms of a training step -
    profiles/HISTORY.md round 6; tools/train_segment_calls.py lists the call sites), a single segment over a trunk layer's 1.1 M rows 0.9 ms.  So: COLSUM_CHUNK-row
    chunks, level after level, until at most 2 * COLSUM_CHUNK rows are left (1.1 M rows of 64 channels: viewed as 276 k rows
    of 256, 2 156 -> 17 -> 1)."""
    rows, C = int(X.shape[0]), int(X.shape[1])
    cache = eng.__dict__.setdefault('_colsum_segs', {})  # the segment tables per row count: small uploads saved per call
    # narrow rows: a lane of the kernel owns 4 channels, so a 64-channel tensor would keep 16 lanes of a wave busy - k
    # consecutive rows are viewed as one row of k * C channels (the sums of the rows = r mod k classes), folded at the end
    fold, C0 = 1, C
    if X.is_contiguous():
        while C * 2 <= 256 and rows % 2 == 0 and rows > 2 * COLSUM_CHUNK:
            fold, C, rows = fold * 2, C * 2, rows // 2
        X = X.view(rows, C)
    while True:
        key = (rows, str(X.device))
        seg = cache.get(key)
        if seg is None:
            if len(cache) > 256:
                cache.clear()
            if rows > 2 * COLSUM_CHUNK:
                n = -(-rows // COLSUM_CHUNK)
                start = np.arange(n) * COLSUM_CHUNK
                seg = Segments(start, np.minimum(COLSUM_CHUNK, rows - start), np.ones(n), np.zeros(n), X.device, div=np.ones(n))
            else:
                seg = Segments([0], [rows], [1], [0], X.device, div=[1])
            cache[key] = seg
        out = torch.empty(seg.n, C, dtype=torch.float32, device=X.device)
        eng.ops.segment_mean(X, C, seg, out, use_group=False)
        if seg.n == 1:
            return out[0] if fold == 1 else out.view(fold, C0).sum(0)
        X, rows = out, seg.n


def _gn_backward(eng, plan, L, dA, out=None, relu=True):
    """dA = gradient w.r.t. relu(GroupNorm(Y)) (or GroupNorm(Y) when ``relu`` is False) -> (dY, dgamma [C],
    dbeta [C]).  ``out``: view to write dY into."""
    ops, tiles, C, dev = eng.ops, L.tiles, L.C, dA.device
    P = torch.empty(tiles.T, 2, C, dtype=torch.float32, device=dev)
    ops.gn_bwd_partial(dA, L.Y, C, L.sc1, L.sh1, L.gamma, L.beta, relu, tiles, P)
    seg_g, seg_t = _aux(plan).tile_sums(tiles, dev)
    S = torch.empty(tiles.G * 2, C, dtype=torch.float32, device=dev)
    tot = torch.empty(2, C, dtype=torch.float32, device=dev)
    P2 = P.view(2 * tiles.T, C)
    _segsum(eng, P2, C, seg_g, S)
    _segsum(eng, P2, C, seg_t, tot)
    M = torch.empty(tiles.G, 2, C, dtype=torch.float32, device=dev)
    ops.gn_bwd_finalize(S, tiles, C, L.NG, L.gamma, M)
    dY = out if out is not None else torch.empty(tiles.R, C, dtype=torch.float32, device=dev)
    ops.gn_bwd_apply(dA, L.Y, C, L.sc1, L.sh1, L.gamma, L.beta, relu, M, tiles, dY)
    return dY, tot[1], tot[0]


def affinity_backward(eng, plan, F, t, d_link, d_new, d_end):
    """Returns (dF [nR, Lt, 512], {reference state_dict key: gradient tensor in the parameter's shape})."""
    ops, lk, PT, VT = eng.ops, eng.P['w_link'], plan.pair_tiles, plan.v_tiles
    nR, Lt, R = plan.nR, plan.Lt, plan.pair_tiles.R
    dev = F.device
    aux = _aux(plan)
    Ff = F.reshape(nR * Lt, 512)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    g = {}

    def weight_grad(name, bname, dY, tiles, N, K, **kw):
        # the reduction over the rows is split into contiguous shares of the tiles (one workgroup walks one share
        # for one 64 x 64 tile of dW); the partial sums are added by the strided-sum kernel: deterministic
        ns = max(1, min(16, tiles.T // 8))
        dWp, dbp = new(ns, N * K), new(ns, N)
        ops.gemm_tn(dY, tiles, N, K, dWp, dbp, nsplit=ns, **kw)
        g[name], g[bname] = (_colsum(eng, dWp).view(N, K), _colsum(eng, dbp)) if ns > 1 else (dWp.view(N, K), dbp.view(N))

    # ---- link branch, from the scores back to dYa[:, 512:] ----
    d_link = d_link.reshape(-1).contiguous()
    if eng.softmax_mode != 'none':
        dlogits = new(R)
        ops.softmax_pairs_bwd(t['logits'], d_link, dlogits, PT.g_row0, plan.pg_N, plan.pg_M, PT.G, plan.max_nm,
                              SOFTMAX_MODES[eng.softmax_mode])
    else:
        dlogits = d_link
    L7, L4, L1, NE0, V1, V4 = t['l7'], t['l4'], t['l1'], t['ne0'], t['v1'], t['v4']
    dA6, PW = new(R, 128), new(PT.T, 132)
    ops.rowdot_bwd(L7.Y, 128, lk['w9'], lk['b9'], L7.sc, L7.sh, PT, ACT_NONE, dlogits, None, dA6, PW)
    pw = _colsum(eng, PW)
    g['w9'], g['b9'] = pw[:128], pw[128:129]
    dY6, g['g7'], g['be7'] = _gn_backward(eng, plan, L7, dA6)
    weight_grad('w6', 'b6', dY6, PT, 128, 512, X=L4.Y, sc=L4.sc, sh=L4.sh, amode=A_NORM_RELU)
    dA3 = new(R, 512)
    dgrad_gemm(eng, lk['w6'], PT, dY6, dA3)
    dY3, g['g4'], g['be4'] = _gn_backward(eng, plan, L4, dA3)
    weight_grad('w3', 'b3', dY3, PT, 512, 512, X=L1.Y, sc=L1.sc, sh=L1.sh, amode=A_NORM_RELU)
    dA1 = new(R, 512)
    dgrad_gemm(eng, lk['w3'], PT, dY3, dA1)
    dYa = new(R, 1024)
    _, g['g1'], g['be1'] = _gn_backward(eng, plan, L1, dA1, out=dYa[:, 512:1024])
    # ---- new / end branch, from the scores back to dYa[:, :512] ----
    d_ne = torch.stack([d_new, d_end]).reshape(-1).contiguous()
    dAv1, PW = new(VT.R, 128), new(VT.T, 132)
    ops.rowdot_bwd(V4.Y, 128, lk['nw6'], lk['nb6'], V4.sc, V4.sh, VT, ACT_SIGMOID, d_ne, plan.v_omap, dAv1, PW)
    pw = _colsum(eng, PW)
    g['nw6'], g['nb6'] = pw[:128], pw[128:129]
    dYv1, g['ng4'], g['nbe4'] = _gn_backward(eng, plan, V4, dAv1)
    weight_grad('nw3', 'nb3', dYv1, VT, 128, 512, X=V1.Y, sc=V1.sc, sh=V1.sh, amode=A_NORM_RELU)
    dAv0 = new(VT.R, 512)
    dgrad_gemm(eng, lk['nw3'], VT, dYv1, dAv0)
    dYv0, g['ng1'], g['nbe1'] = _gn_backward(eng, plan, V1, dAv0)
    weight_grad('nw0', 'nb0', dYv0, VT, 512, 512, X=t['V'], amode=A_PLAIN)
    dV = new(VT.R, 512)
    dgrad_gemm(eng, lk['nw0'], VT, dYv0, dV)
    dAne = new(R, 512)
    ops.pair_expand_bwd(dV, dAne, 512, PT, PT.g_row0, plan.pg_N, plan.pg_M, aux.vrow0)
    _, g['g_ne0'], g['be_ne0'] = _gn_backward(eng, plan, NE0, dAne, out=dYa[:, 0:512])
    # ---- the stacked first layer over the pairwise tensor, and the pairwise operand generation ----
    weight_grad('wa', 'ba', dYa, PT, 1024, 512, FA=Ff, FB=Ff, pair=t['pair'], amode=A_PAIR, pairop=t['pairop'])
    dWa, dba = g.pop('wa'), g.pop('ba')
    dX = new(R, 512)
    dgrad_gemm(eng, lk['wa'], PT, dYa, dX)
    dF = torch.zeros(nR * Lt, 512, dtype=torch.float32, device=dev)
    common = (PT.g_row0, plan.pg_N, plan.pg_M, plan.pg_aoff, plan.pg_boff)
    ops.pair_bwd(dX, Ff, dF, 512, *common, aux.a_grp, aux.a_idx, t['pairop'], 0)
    ops.pair_bwd(dX, Ff, dF, 512, *common, aux.b_grp, aux.b_idx, t['pairop'], 1)
    # ---- reference-keyed parameter gradients (flat / 2-D; the caller reshapes to the parameter's shape) ----
    out = {'w_link.w_new_end.conv0.0.weight': dWa[0:512], 'w_link.w_new_end.conv0.0.bias': dba[0:512],
           'w_link.conv1.0.weight': dWa[512:1024], 'w_link.conv1.0.bias': dba[512:1024]}
    for name, key in PARAM_KEYS.items():
        out[key] = g[name]
    return dF.view(nR, Lt, 512), out


# ======================================================================================================================
# Second slice: the rest of the head - fusion module A / B / C and the negative-rejection head w_det in TRAINING mode
# (reference modules/fusion_net.py:31-42,62-70,85-92; tracking_net.py:91-100,149-163 with self.training: BatchNorm1d on
# batch statistics over the 3 modality rows x L detections, no sigmoid, no neg_threshold mask).  Together with the
# pairwise block this is everything between the encoder features `cat` [Lt][1024] and the four score tensors.
# ======================================================================================================================
def _t2(p):
    """conv weight [N, K, 1(, 1)] -> [N][K] fp32 view on the parameter's device"""
    return p.detach().reshape(p.shape[0], -1).contiguous()


def fusion_forward_train(eng, plan, cat):
    """Engine.fuse with the tape: cat [Lt][1024] -> F [3][Lt][512] (GroupNorm only: identical to eval)."""
    ops, fu, D, Lt = eng.ops, eng.P['fusion'], plan.det_tiles, plan.Lt
    dev = cat.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    F = new(3, Lt, 512)
    from .ops import FUSION_MODES
    mode = FUSION_MODES[eng.fusion]
    t = dict(mode=eng.fusion)
    if eng.fusion == 'A':
        y0, part = new(Lt, 512), new(D.T, 2, 512)
        eng._gemm(fu, 'w0', D, 512, 1024, X=cat, bias=fu['b0'], Y=y0, part=part)
        L0 = _norm_layer(eng, part, D, y0, 512, 512, fu['g0'], fu['be0'])
        ops.fusion_combine(mode, cat, y0, None, L0.sc, L0.sh, None, None, D, F, Lt, 512)
        t['L'] = [L0]
        return F, t
    N = 512 if eng.fusion == 'B' else 1024
    Ls, ys = [], []
    for j, x in enumerate((cat[:, 0:512], cat[:, 512:1024])):
        y, part = new(Lt, N), new(D.T, 2, N)
        eng._gemm(fu, 'w%d' % j, D, N, 512, X=x, bias=fu['b%d' % j], Y=y, part=part)
        # the GroupNorm acts on the `input` half only (fusion C: columns 512.. of the stacked [gate ; input] layer)
        Ls.append(_norm_layer(eng, part[:, :, N - 512:], D, y[:, N - 512:], 512, 512, fu['g%d' % j][N - 512:],
                              fu['be%d' % j][N - 512:]))
        ys.append(y)
    ops.fusion_combine(mode, cat, ys[0], ys[1], Ls[0].sc, Ls[0].sh, Ls[1].sc, Ls[1].sh, D, F, Lt, 512)
    t['L'], t['ys'] = Ls, ys
    return F, t


def fusion_backward(eng, plan, cat, t, dF):
    """dF [3][Lt][512] -> (dcat [Lt][1024], {fusion_module.* key: grad})."""
    ops, fu, D, Lt = eng.ops, eng.P['fusion'], plan.det_tiles, plan.Lt
    dev = cat.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    fm = 'fusion_module.'
    out = {}
    dcat = new(Lt, 1024)

    def wgrad(dY, N, K, X):
        dW, db = new(1, N * K), new(1, N)
        ops.gemm_tn(dY, D, N, K, dW, db, X=X, amode=A_PLAIN)
        return dW.view(N, K), db.view(N)

    if t['mode'] == 'A':
        L0 = t['L'][0]
        dy0, out[fm + 'input_w.1.weight'], out[fm + 'input_w.1.bias'] = _gn_backward(eng, plan, L0, dF[2], relu=False)
        out[fm + 'input_w.0.weight'], out[fm + 'input_w.0.bias'] = wgrad(dy0, 512, 1024, cat)
        dconv = new(Lt, 1024)
        dgrad_gemm(eng, fu['w0'], D, dy0, dconv)
        ops.add_rows(dconv[:, 0:512], dF[0], dcat[:, 0:512], 512)
        ops.add_rows(dconv[:, 512:1024], dF[1], dcat[:, 512:1024], 512)
        return dcat, out
    names = (('gate_p', 'input_p'), ('gate_i', 'input_i'))  # *_p acts on the IMAGE half (naming trap)
    if t['mode'] == 'B':
        for j in range(2):
            Lj = t['L'][j]
            dy, out[fm + names[j][1] + '.1.weight'], out[fm + names[j][1] + '.1.bias'] = _gn_backward(
                eng, plan, Lj, dF[2], relu=False)
            out[fm + names[j][1] + '.0.weight'], out[fm + names[j][1] + '.0.bias'] = wgrad(
                dy, 512, 512, cat[:, 512 * j:512 * (j + 1)])
            dconv = new(Lt, 512)
            dgrad_gemm(eng, fu['w%d' % j], D, dy, dconv)
            ops.add_rows(dconv, dF[j], dcat[:, 512 * j:512 * (j + 1)], 512)
        return dcat, out
    # C: gates + normalised inputs
    L0, L1 = t['L']
    y0, y1 = t['ys']
    DY = [new(Lt, 1024), new(Lt, 1024)]
    DN = [new(Lt, 512), new(Lt, 512)]
    ops.fusion_c_bwd(dF[2], y0, y1, L0.sc, L0.sh, L1.sc, L1.sh, D, DY[0], DY[1], DN[0], DN[1], 512)
    for j in range(2):
        Lj = t['L'][j]
        _, out[fm + names[j][1] + '.1.weight'], out[fm + names[j][1] + '.1.bias'] = _gn_backward(
            eng, plan, Lj, DN[j], out=DY[j][:, 512:1024], relu=False)
        dW, db = wgrad(DY[j], 1024, 512, cat[:, 512 * j:512 * (j + 1)])
        out[fm + names[j][0] + '.0.weight'], out[fm + names[j][0] + '.0.bias'] = dW[0:512], db[0:512]
        out[fm + names[j][1] + '.0.weight'], out[fm + names[j][1] + '.0.bias'] = dW[512:1024], db[512:1024]
        dconv = new(Lt, 512)
        dgrad_gemm(eng, fu['w%d' % j], D, DY[j], dconv)
        ops.add_rows(dconv, dF[j], dcat[:, 512 * j:512 * (j + 1)], 512)
    return dcat, out


def det_forward_train(eng, model, plan, F):
    """w_det in TRAINING mode (tracking_net.py:149-151 with self.training): conv -> BatchNorm1d on the batch statistics of
    the 3 x L positions -> ReLU, twice, then the 1-channel conv; raw scores (no sigmoid, no threshold mask).
    F [nR][Lt][512] -> det [nR][Lt].  One sample per call, like the reference (its batch is 1: the BatchNorm statistics
    are those of one sample)."""
    if plan.B != 1:
        raise NotImplementedError('training-mode w_det: one sample per call (BatchNorm statistics are per forward)')
    ops, T = eng.ops, plan.F_tiles
    R = plan.nR * plan.Lt
    dev = F.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    wd = model.w_det
    X = F.reshape(R, 512)
    t = {}
    h0, part = new(R, 512), new(T.T, 2, 512)
    ops.gemm(_t2(wd[0].weight), T, 512, 512, X=X, bias=wd[0].bias.detach(), Y=h0, part=part)
    t['d0'] = _norm_layer(eng, part, T, h0, 512, 512, wd[1].weight.detach(), wd[1].bias.detach())
    h1, part = new(R, 256), new(T.T, 2, 256)
    ops.gemm(_t2(wd[3].weight), T, 256, 512, X=h0, bias=wd[3].bias.detach(), Y=h1, part=part, sc=t['d0'].sc,
             sh=t['d0'].sh, amode=A_NORM_RELU)
    t['d1'] = _norm_layer(eng, part, T, h1, 256, 256, wd[4].weight.detach(), wd[4].bias.detach())
    det = new(plan.nR, plan.Lt)
    t['w6'], t['b6'] = wd[6].weight.detach().reshape(-1).contiguous(), float(wd[6].bias.item())
    ops.rowdot(h1, 256, t['w6'], t['b6'], T, det.view(-1), sc=t['d1'].sc, sh=t['d1'].sh, act=ACT_NONE)
    return det, t


def det_batch_stats(layer, n):
    """(mean, unbiased variance) of a training-mode BatchNorm layer from its unit statistics - what
    ``running_mean`` / ``running_var`` are updated with (momentum update left to the caller)."""
    rstd = layer.sc1[0]
    mean = -layer.sh1[0] / rstd
    var = (1.0 / (rstd * rstd) - EPS) * (n / max(n - 1, 1))
    return mean, var


def det_backward(eng, model, plan, F, t, d_det):
    """d_det [nR][Lt] -> (dF [nR][Lt][512], {w_det.* key: grad})."""
    ops, T = eng.ops, plan.F_tiles
    R = plan.nR * plan.Lt
    dev = F.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    wd = model.w_det
    X = F.reshape(R, 512)
    out = {}
    D0, D1 = t['d0'], t['d1']
    dA1, PW = new(R, 256), new(T.T, 260)
    ops.rowdot_bwd(D1.Y, 256, t['w6'], t['b6'], D1.sc, D1.sh, T, ACT_NONE, d_det.reshape(-1).contiguous(), None, dA1, PW)
    pw = _colsum(eng, PW)
    out['w_det.6.weight'], out['w_det.6.bias'] = pw[:256], pw[256:257]
    dY1, out['w_det.4.weight'], out['w_det.4.bias'] = _gn_backward(eng, plan, D1, dA1)
    dW, db = new(1, 256 * 512), new(1, 256)
    ops.gemm_tn(dY1, T, 256, 512, dW, db, X=D0.Y, sc=D0.sc, sh=D0.sh, amode=A_NORM_RELU)
    out['w_det.3.weight'], out['w_det.3.bias'] = dW.view(256, 512), db.view(256)
    dA0 = new(R, 512)
    dgrad_gemm(eng, _t2(wd[3].weight), T, dY1, dA0)
    dY0, out['w_det.1.weight'], out['w_det.1.bias'] = _gn_backward(eng, plan, D0, dA0)
    dW, db = new(1, 512 * 512), new(1, 512)
    ops.gemm_tn(dY0, T, 512, 512, dW, db, X=X, amode=A_PLAIN)
    out['w_det.0.weight'], out['w_det.0.bias'] = dW.view(512, 512), db.view(512)
    dF = new(R, 512)
    dgrad_gemm(eng, _t2(wd[0].weight), T, dY0, dF)
    return dF.view(plan.nR, plan.Lt, 512), out


def head_forward_train(eng, model, plan, cat):
    """cat [Lt][1024] (encoder features) -> (det, link, new, end, tape): fusion -> w_det (training mode) + pairwise block"""
    F, tf = fusion_forward_train(eng, plan, cat)
    det, td = det_forward_train(eng, model, plan, F)
    link, new, end, ta = affinity_forward_train(eng, plan, F)
    return det, link, new, end, dict(F=F, fusion=tf, det=td, aff=ta)


def head_backward(eng, model, plan, cat, tape, d_det, d_link, d_new, d_end):
    """-> (dcat [Lt][1024], {state_dict key: grad} for fusion_module.*, w_det.*, w_link.*)"""
    F = tape['F']
    dF_a, grads = affinity_backward(eng, plan, F, tape['aff'], d_link, d_new, d_end)
    dF_d, g_d = det_backward(eng, model, plan, F, tape['det'], d_det)
    dF = torch.empty_like(dF_a)
    eng.ops.add_rows(dF_a.view(-1, 512), dF_d.view(-1, 512), dF.view(-1, 512), 512)
    dcat, g_f = fusion_backward(eng, plan, cat, tape['fusion'], dF)
    grads.update(g_d)
    grads.update(g_f)
    return dcat, grads


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cat, eng, model, plan, keys, *params):
        c = cat.detach().contiguous()
        with eng.fp32_mlp():
            det, link, new, end, tape = head_forward_train(eng, model, plan, c)
        eng._last_head_tape = tape  # head_autograd reads the batch statistics from here (grad mode or not)
        ctx.eng, ctx.model, ctx.plan, ctx.tape, ctx.keys = eng, model, plan, tape, keys
        ctx.save_for_backward(cat)
        ctx.shapes = [tuple(p.shape) for p in params]
        return det, link, new, end

    @staticmethod
    def backward(ctx, d_det, d_link, d_new, d_end):
        (cat,) = ctx.saved_tensors
        pl = ctx.plan
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=cat.device)
        d_det = d_det if d_det is not None else z(pl.nR, pl.Lt)
        d_link = d_link if d_link is not None else z(pl.pair_tiles.R)
        d_new = d_new if d_new is not None else z(pl.nR, pl.Lt)
        d_end = d_end if d_end is not None else z(pl.nR, pl.Lt)
        dcat, grads = head_backward(ctx.eng, ctx.model, pl, cat.detach().contiguous(), ctx.tape, d_det.contiguous(),
                                    d_link.contiguous(), d_new.contiguous(), d_end.contiguous())
        pg = tuple(grads[k].reshape(s) for k, s in zip(ctx.keys, ctx.shapes))
        return (dcat, None, None, None, None) + pg


def head_autograd(model, plan, cat, update_running_stats=True):
    """Differentiable head of ``model`` in training mode: encoder features cat [Lt][1024] (image | LiDAR) ->
    (det [nR, Lt] raw scores, link flat, new [nR, Lt], end [nR, Lt]); ``backward()`` fills ``cat.grad`` and the ``.grad``
    of every fusion_module / w_det / w_link parameter.  With ``update_running_stats`` the BatchNorm buffers of w_det take
    the momentum update PyTorch's training mode does."""
    eng = _current_engine(model)
    named = [(k, p) for k, p in model.named_parameters() if k.split('.')[0] in ('fusion_module', 'w_det', 'w_link')]
    keys = tuple(k for k, _ in named)
    eng._last_head_tape = None
    out = _HeadFn.apply(cat, eng, model, plan, keys, *[p for _, p in named])
    if update_running_stats:
        # like nn.BatchNorm1d in training mode, the buffers take the momentum update whether or not autograd records
        # (the tape is handed over through the engine, not through grad_fn, which is None under no_grad)
        tape, eng._last_head_tape = eng._last_head_tape, None
        if tape is not None:
            n = plan.nR * plan.Lt
            with torch.no_grad():
                for idx, name in ((1, 'd0'), (4, 'd1')):
                    bn = model.w_det[idx]
                    mean, var = det_batch_stats(tape['det'][name], n)
                    m = bn.momentum if bn.momentum is not None else 0.1
                    bn.running_mean.mul_(1 - m).add_(m * mean)
                    bn.running_var.mul_(1 - m).add_(m * var)
                    bn.num_batches_tracked += 1
    return out


def _current_engine(model):
    """The model's engine with the head packed from the parameters as they are NOW: fusion and w_link run from the
    packed copies, w_det's training forward from the live parameters - after an ``optimizer.step()`` (or any in-place
    edit: the parameters' version counters moved) the head is re-packed first, so forward and gradients never mix two
    generations of weights."""
    eng = model.engine()
    if hasattr(model, 'head_is_current') and not model.head_is_current():
        # on the device (no host packing; the fp16-split copies go stale and the training forward uses fp32 weights)
        eng = model.refresh_head_device() if hasattr(model, 'refresh_head_device') else model.refresh_head()
    return eng


class _AffinityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, eng, plan, keys, *params):
        with eng.fp32_mlp():
            link, new, end, tape = affinity_forward_train(eng, plan, F.detach().contiguous())
        ctx.eng, ctx.plan, ctx.tape, ctx.keys = eng, plan, tape, keys
        ctx.save_for_backward(F)
        ctx.shapes = [tuple(p.shape) for p in params]
        return link, new, end

    @staticmethod
    def backward(ctx, d_link, d_new, d_end):
        (F,) = ctx.saved_tensors
        zeros = lambda ref, shape: torch.zeros(shape, dtype=torch.float32, device=F.device)
        nR, Lt = ctx.plan.nR, ctx.plan.Lt
        d_link = d_link if d_link is not None else zeros(F, (ctx.plan.pair_tiles.R,))
        d_new = d_new if d_new is not None else zeros(F, (nR, Lt))
        d_end = d_end if d_end is not None else zeros(F, (nR, Lt))
        dF, grads = affinity_backward(ctx.eng, ctx.plan, F.detach().contiguous(), ctx.tape, d_link.contiguous(),
                                      d_new.contiguous(), d_end.contiguous())
        pg = tuple(grads[k].reshape(s) for k, s in zip(ctx.keys, ctx.shapes))
        return (dF, None, None, None) + pg


def affinity_autograd(model, plan, F):
    """Differentiable pairwise block of ``model`` (a TrackingNet on the device): F [nR, Lt, 512] ->
    (link flat [sum nR*N*M], new [nR, Lt], end [nR, Lt]) attached to the autograd graph; ``backward()`` fills
    ``F.grad`` and the ``.grad`` of every ``model.w_link`` parameter.  The packed weights are those of
    ``model.engine()``, re-packed here when a head parameter changed since (``model.head_is_current()``)."""
    eng = _current_engine(model)
    named = [(k, p) for k, p in model.named_parameters() if k.startswith('w_link.')]
    keys = tuple(k for k, _ in named)
    return _AffinityFn.apply(F, eng, plan, keys, *[p for _, p in named])
