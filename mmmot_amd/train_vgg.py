"""Training step, third slice: the image encoder in TRAINING mode, differentiable.

``appearance_autograd(model, plan, crops)`` is ``AppearanceNet.forward`` (reference modules/appear_net.py:166-190 with the
VGG16-BN stages of modules/vgg.py:67-80) under ``model.train()``: every BatchNorm2d normalises with the statistics of the
batch - the L crops of the one sample of a step, all pixels - and updates its running buffers; the SkipPool heads are
GroupNorm(1, .) only.  Returns the L x 512 features on the autograd graph of every ``appearance.*`` parameter (convolution
weights and biases, BatchNorm affines, head parameters).

The inference trunk cannot be reused: it folds the running statistics into fp16-split weights, fuses ReLU / pooling into
the convolution epilogue and never stores a pre-BatchNorm tensor.  This path is separate and plain (fp32 NHWC, exact fp32
matrix cores): per layer ``mmmot_conv3x3_raw`` -> ``mmmot_rows_stats`` + ``mmmot_gn_finalize`` (BatchNorm over the batch IS
the per-channel GroupNorm with one group) -> ``mmmot_bn_relu_pool``; backward ``mmmot_maxpool_bwd`` -> the GroupNorm
backward kernels of csrc/backward.hip -> ``mmmot_conv3x3_wgrad`` (dW) and ``mmmot_conv3x3_raw`` on the flipped / transposed
weights (dX).  Everything numeric is in libmmmot_hip.so; torch permutes weights (data movement) and keeps the graph.
DropBlock (``dropblock`` > 0; every shipped config sets 0): the seed masks are drawn on the host like the reference's, the
scaling of the stage outputs and of their gradients runs on the device (_dropblock_scale).
"""
import numpy as np
import torch

from .backward import _colsum, _gn_backward, _norm_layer, det_batch_stats, dgrad_gemm
from .ops import ACT_NONE, ACT_RELU, A_NORM_RELU, A_PLAIN
from .pack import VGG_STAGES
from .plan import RowTiles, Segments
from .train import _weight_grad

EPS = 1e-5
HEAD_KEYS = ('0.weight', '0.bias', '1.weight', '1.bias', '2.weight', '2.bias', '4.weight', '4.bias', '5.weight', '5.bias')


def _tiles(cache, rows, dev, each=None):
    """row tiling of a [rows][C] tensor as ONE normalisation group (BatchNorm over the batch), or (each=1) one group per
    row (GroupNorm(1, C) on an L x C x 1 x 1 tensor: a LayerNorm per detection)"""
    key = (rows, each)
    if key not in cache:
        cache[key] = RowTiles([1] * rows if each else [rows], dev)
    return cache[key]


def _ident_rows(cache, n, dev):
    key = ('ident', n)
    if key not in cache:
        cache[key] = torch.arange(n, dtype=torch.int32, device=dev)
    return cache[key]


def _colsum_big(eng, X):
    """column sums of a tall [rows][C] tensor (the bias gradient of a trunk layer): _colsum's levels of 128-row chunks"""
    return _colsum(eng, X)


def _f16_convs(ops):
    """training-mode trunk convolutions (forward and input gradient) on the fp16 matrix cores: the product backend in its
    default arithmetic (MMMOT_GEMM_TN=f32 keeps the exact fp32 MFMA everywhere; the torch emulation of the tests has no such
    entry point and runs the fp32 statement)"""
    return bool(getattr(ops, 'tn_f16', False)) and hasattr(ops, 'conv3x3_raw_hl16')


def _conv_params(model):
    """[(stage, conv index inside the stage's Sequential, bn module, cin, cout, pool, last-of-stage)] in execution order"""
    out = []
    for s, stage in enumerate(VGG_STAGES):
        for (idx, cin, cout, pool) in stage:
            seq = model.appearance.layers[s]
            out.append((s, idx, seq[idx + 1], cin, cout, pool, idx == stage[-1][0]))
    return out


def _param_list(model):
    """(keys, tensors) of every appearance.* parameter, in a fixed order"""
    keys, ts = [], []
    for k, p in model.appearance.named_parameters():
        keys.append(k)
        ts.append(p)
    return keys, ts


def appearance_forward_train(eng, model, plan, crops, P):
    """crops [L][3][S][S]; P: {appearance-relative key: detached fp32 tensor}.  Returns (feats [L][512], tape)."""
    ops, L, S = eng.ops, plan.Lt, plan.S
    dev = crops.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    if plan.B != 1:
        raise NotImplementedError('training-mode trunk: one sample per call (BatchNorm2d statistics are those of the forward)')
    dropblock = int(getattr(model.appearance, 'dropblock', 0) or 0)  # block size; stages 2 / 3 (appear_net.py:143-152)
    cache = plan.__dict__.setdefault('_vgg_train_tiles', {})
    f16 = _f16_convs(ops)
    feats = new(L, 512)
    tape = dict(layers=[], heads=[])
    x, H, W = crops.contiguous(), S, S
    first = True
    for li, (s, cidx, bn, cin, cout, pool, last) in enumerate(_conv_params(model)):
        pre = 'layers.%d.' % s
        w = P[pre + '%d.weight' % cidx]
        b = P[pre + '%d.bias' % cidx]
        if first:  # [Cout][3][3][3] (n, c, ky, kx) -> [Cout][k = (ky*3+kx)*3 + c], padded to 32
            wp = torch.zeros(cout, 32, dtype=torch.float32, device=dev)
            wp[:, :27] = w.permute(0, 2, 3, 1).reshape(cout, 27)
        else:      # -> [tap][Cout][Cin]
            wp = w.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous()
        rows = L * H * W
        Z = new(rows, cout)
        amw = None
        if f16 and not first:
            # f16x3 (round 4): activations split once (hl16), weights scaled by a device-side power of two (max |w| ->
            # [2^13, 2^14): their lo halves stay normal fp16 numbers) and split, the trunk kernel of the inference path with
            # a raw fp32 output.  Nothing crosses the host: the step's weights are whatever the optimizer left on the device.
            x16, w16, amw, osc = new(x.shape[0], cin), torch.empty_like(wp), new(1), new(cout)
            ops.hl16_pack(x, x16)
            ops.absmax(wp, amw)
            ops.hl16_pack_pow2(wp, w16, amw, 14)
            ops.pow2_oscale(osc, amw, 14)
            ops.conv3x3_raw_hl16(x16, w16, b, Z, L, H, W, cin, cout, osc)
        else:
            ops.conv3x3_raw(x, wp, b, Z, L, H, W, cin, cout, first)
        T = _tiles(cache, rows, dev)
        part = new(T.T, 2, cout)
        ops.rows_stats(Z, cout, T, part)
        Lyr = _norm_layer(eng, part, T, Z, cout, cout, P[pre + '%d.weight' % (cidx + 1)], P[pre + '%d.bias' % (cidx + 1)])
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        A = new(L * Ho * Wo, cout)
        ops.bn_relu_pool(Z, cout, Lyr.sc, Lyr.sh, L, H, W, pool, A)
        tape['layers'].append(dict(L=Lyr, x=x, wp=wp, H=H, W=W, cin=cin, cout=cout, pool=pool, first=first, stage=s,
                                   last=last, cidx=cidx, bn=bn, rows=rows, amw=amw))
        x, H, W, first = A, Ho, Wo, False
        if last:
            tape['heads'].append(_head_forward(eng, plan, cache, s, x, H * W, cout, P, feats,
                                               drop=_dropblock_scale(L, H, W, dropblock, dev) if (dropblock and s >= 2) else None))
    return feats, tape


DROPBLOCK_PROB = 0.1  # reference modules/dropblock.py:22: DropBlock2D's default (appear_net.py:18 passes the block size only)


def _dropblock_scale(L, H, W, block_size, dev):
    """DropBlock2D.forward in training mode (reference modules/dropblock.py:28-68) as ONE factor per (crop, pixel) row:
    block_mask * numel / sum.  The Bernoulli seed mask is drawn on the HOST with the global torch generator, exactly like
    the reference (`torch.rand(N, H, W)` then `.to(device)`): under the same `torch.manual_seed` and the same order of
    draws (stage 2, then stage 3) the masks ARE the reference's.  The block growth (a max pool over L x H x W <= a few
    thousand values) is mask bookkeeping, done where the mask is drawn; the tensor work - scaling the stage output before
    the average pool and its gradient behind it - runs on the device (mmmot_rows_gather_scale)."""
    import torch.nn.functional as Fn
    gamma = DROPBLOCK_PROB / (block_size ** 2)
    mask = (torch.rand(L, H, W) < gamma).float()
    bm = Fn.max_pool2d(mask[:, None], kernel_size=(block_size, block_size), stride=(1, 1), padding=block_size // 2)
    if block_size % 2 == 0:
        bm = bm[:, :, :-1, :-1]
    bm = 1 - bm.squeeze(1)
    scale = bm * (bm.numel() / bm.sum())   # out = x * block_mask * numel / sum (dropblock.py:50-53), same fp32 operations
    return scale.reshape(-1).contiguous().to(dev)


def _head_forward(eng, plan, cache, s, x, hw, C, P, feats, drop=None):
    """SkipPool (appear_net.py:9-32) of stage s on its output x [L * hw][C] -> feats[:, 128 s : 128 (s + 1)], with tape.
    ``drop``: the DropBlock factor per pixel row (_dropblock_scale), applied in front of the average pool."""
    ops, L = eng.ops, plan.Lt
    dev = x.device
    new = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
    pre = 'global_pool.%d.fc.' % s
    T1 = _tiles(cache, L, dev, each=1)
    key = ('pool', L, hw)  # cached with the plan: a Segments table is five small host-to-device copies (4.5 ms per step
    if key not in cache:   # over the four heads, cProfile round 4)
        if hw > 512:
            # a crop's map in 128-row chunks (sums), then the chunks of a crop (divided by hw): 22 workgroups walking
            # 3 136 rows each took 0.2 - 0.3 ms
            nch = -(-hw // 128)
            k = np.arange(nch)
            s1 = (np.arange(L)[:, None] * hw + k[None] * 128).reshape(-1)
            c1 = np.tile(np.minimum(128, hw - k * 128), L)
            cache[key] = (Segments(s1, c1, np.ones(len(s1)), np.zeros(len(s1)), dev, div=np.ones(len(s1))),
                          Segments(np.arange(L) * nch, np.full(L, nch), np.ones(L), np.zeros(L), dev, div=np.full(L, hw)))
        else:
            cache[key] = Segments(np.arange(L) * hw, np.full(L, hw), np.ones(L), np.zeros(L), dev)
    seg = cache[key]
    Pm = new(L, C)
    if drop is not None:                                        # x * block_mask * numel / sum (dropblock.py:50-53)
        ident = _ident_rows(cache, L * hw, dev)
        xd = new(L * hw, C)
        ops.rows_gather_scale(x, ident, drop, xd, C)
        x = xd
    if isinstance(seg, tuple):                                  # AdaptiveAvgPool2d(1)
        chunk_sums = new(seg[0].n, C)
        ops.segment_mean(x, C, seg[0], chunk_sums, use_group=False)
        ops.segment_mean(chunk_sums, C, seg[1], Pm, use_group=False)
    else:
        ops.segment_mean(x, C, seg, Pm, use_group=False)
    part = new(T1.T, 2, C)
    ops.rows_stats(Pm, C, T1, part)
    L0 = _norm_layer(eng, part, T1, Pm, C, 1, P[pre + '0.weight'], P[pre + '0.bias'])
    x0 = new(L, C)
    ops.affine_act(Pm, C, L0.sc, L0.sh, T1, ACT_NONE, x0)      # GroupNorm(1, C), no ReLU
    w1 = P[pre + '1.weight'].flatten(1).contiguous()
    C4 = w1.shape[0]
    h1, part = new(L, C4), new(T1.T, 2, C4)
    ops.gemm(w1, T1, C4, C, X=x0, bias=P[pre + '1.bias'], Y=h1, part=part)
    L2 = _norm_layer(eng, part, T1, h1, C4, 1, P[pre + '2.weight'], P[pre + '2.bias'])
    w4 = P[pre + '4.weight'].flatten(1).contiguous()
    h2, part = new(L, 128), new(T1.T, 2, 128)
    ops.gemm(w4, T1, 128, C4, X=h1, bias=P[pre + '4.bias'], Y=h2, part=part, sc=L2.sc, sh=L2.sh, amode=A_NORM_RELU)
    L5 = _norm_layer(eng, part, T1, h2, 128, 1, P[pre + '5.weight'], P[pre + '5.bias'])
    ops.affine_act(h2, 128, L5.sc, L5.sh, T1, ACT_RELU, feats[:, 128 * s:128 * (s + 1)])
    return dict(L0=L0, L2=L2, L5=L5, x0=x0, w1=w1, w4=w4, C=C, C4=C4, hw=hw, T1=T1, stage=s, drop=drop)


def _head_backward(eng, plan, hd, dOut, g):
    """dOut [L][128] (a column slice of d feats) -> d(stage output) as per-crop rows [L][C] (to be spread over hw pixels)"""
    ops, L = eng.ops, plan.Lt
    dev = dOut.device
    new = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
    pre = 'global_pool.%d.fc.' % hd['stage']
    T1, C, C4 = hd['T1'], hd['C'], hd['C4']
    dOutc = dOut.contiguous()
    dh2, g[pre + '5.weight'], g[pre + '5.bias'] = _gn_backward(eng, plan, hd['L5'], dOutc)
    L2 = hd['L2']
    g[pre + '4.weight'], g[pre + '4.bias'] = _weight_grad(eng, dh2, T1, 128, C4, X=L2.Y, sc=L2.sc, sh=L2.sh, amode=A_NORM_RELU)
    dA1 = new(L, C4)
    dgrad_gemm(eng, hd['w4'], T1, dh2, dA1)
    dh1, g[pre + '2.weight'], g[pre + '2.bias'] = _gn_backward(eng, plan, L2, dA1)
    g[pre + '1.weight'], g[pre + '1.bias'] = _weight_grad(eng, dh1, T1, C4, C, X=hd['x0'], amode=A_PLAIN)
    dx0 = new(L, C)
    dgrad_gemm(eng, hd['w1'], T1, dh1, dx0)
    dP, g[pre + '0.weight'], g[pre + '0.bias'] = _gn_backward(eng, plan, hd['L0'], dx0, relu=False)
    return dP


def _wgrad_shares(dev, cin, cout, rows, f16):
    """pixel shares of one layer's weight-gradient launch.  f16x3 kernel (csrc/train_vgg.hip): a workgroup owns a
    (128- or 64-channel tile pair, tap row, share) and the whole LDS of a CU (two fit with 64 x 64 tiles) - as many shares
    as give ONE round of workgroups over the chip (288 workgroups on 256 CUs ran two rounds, the second an eighth full),
    each with at least four 64-pixel chunks.  fp32 kernel: 64 x 64 tiles per tap, about a thousand workgroups."""
    if not f16:
        tiles_w = (cout // 64) * (cin // 64) * 9
        return int(max(1, min(64, -(-1024 // tiles_w), rows // 256)))
    tn, tk = (128 if cout % 128 == 0 else 64), (128 if cin % 128 == 0 else 64)
    tiles = (cout // tn) * (cin // tk) * 3
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == 'cuda' else 256
    slots = n_cu * (2 if tn == 64 and tk == 64 else 1)
    return int(max(1, min(256, slots // tiles, rows // 256)))


def appearance_backward(eng, model, plan, crops, tape, dF):
    """dF [L][512] -> {appearance-relative key: gradient}"""
    ops, L = eng.ops, plan.Lt
    dev = dF.device
    new = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
    g = {}
    heads = {hd['stage']: hd for hd in tape['heads']}
    dA = None  # gradient w.r.t. the current layer's OUTPUT (post BatchNorm / ReLU / pool), [L * Ho * Wo][Cout]
    for ly in reversed(tape['layers']):
        Lyr, H, W, cin, cout, pool = ly['L'], ly['H'], ly['W'], ly['cin'], ly['cout'], ly['pool']
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        pre = 'layers.%d.' % ly['stage']
        if ly['last']:
            hd = heads[ly['stage']]
            dP = _head_backward(eng, plan, hd, dF[:, 128 * ly['stage']:128 * (ly['stage'] + 1)], g)
            hw = Ho * Wo
            rowidx = torch.arange(L, dtype=torch.int32, device=dev).repeat_interleave(hw)  # pixel row -> crop (data movement)
            scale = torch.full((L,), 1.0 / hw, dtype=torch.float32, device=dev)
            dpool = new(L * hw, cout)
            ops.rows_gather_scale(dP, rowidx, scale, dpool, cout)     # backward of AdaptiveAvgPool2d(1)
            if hd.get('drop') is not None:                            # ... and of the DropBlock factor in front of it
                cache = plan.__dict__.setdefault('_vgg_train_tiles', {})
                dd = new(L * hw, cout)
                ops.rows_gather_scale(dpool, _ident_rows(cache, L * hw, dev), hd['drop'], dd, cout)
                dpool = dd
            if dA is None:
                dA = dpool
            else:
                both = new(L * hw, cout)
                ops.add_rows(dA, dpool, both, cout)
                dA = both
        if pool:
            dApre = new(L * H * W, cout)
            ops.maxpool_bwd(Lyr.Y, cout, Lyr.sc, Lyr.sh, dA, L, H, W, dApre)
        else:
            dApre = dA
        cidx = ly['cidx']
        dZ, g[pre + '%d.weight' % (cidx + 1)], g[pre + '%d.bias' % (cidx + 1)] = _gn_backward(eng, plan, Lyr, dApre)
        g[pre + '%d.bias' % cidx] = _colsum_big(eng, dZ)
        rows = ly['rows']
        if ly['first']:
            nb = max(1, min(1024, rows // 512))
            PW = new(nb, 64 * 28)
            ops.conv3x3_first_wgrad(dZ, ly['x'], L, H, W, PW)
            pw = _colsum(eng, PW).view(64, 28)
            g[pre + '%d.weight' % cidx] = pw[:, :27].reshape(64, 3, 3, 3).permute(0, 3, 1, 2)  # [n][ky][kx][c] -> [n][c][ky][kx]
            dA = None
        else:
            f16 = _f16_convs(ops) and ly['amw'] is not None
            ns = _wgrad_shares(dev, cin, cout, rows, f16)
            dWp = new(ns, 9 * cout * cin)
            if f16:
                amz = new(1)
                ops.absmax(dZ, amz)  # one maximum for both uses of dZ
                ops.conv3x3_wgrad(dZ, ly['x'], L, H, W, cin, cout, ns, dWp, amax=amz)
            else:
                ops.conv3x3_wgrad(dZ, ly['x'], L, H, W, cin, cout, ns, dWp)
            dW = _colsum(eng, dWp) if ns > 1 else dWp[0]
            g[pre + '%d.weight' % cidx] = dW.view(3, 3, cout, cin).permute(2, 3, 0, 1)
            # input gradient: the same convolution kernel on dZ with the taps flipped and Cin / Cout swapped
            wflip = ly['wp'].flip(0).permute(0, 2, 1).contiguous()  # [tap][Cin][Cout]
            zero = torch.zeros(cin, dtype=torch.float32, device=dev)
            dA = new(L * H * W, cin)
            if f16:
                # f16x3: dZ scaled by the power of two that puts its maximum at 2^10 (gradients of 1e-6 sit in fp16's
                # subnormals otherwise - what broke the SGD-step parity of the unscaled attempt of round 3), the weights
                # by theirs; the epilogue's per-channel vector undoes both exactly
                dz16, wf16, osc = new(rows, cout), torch.empty_like(wflip), new(cin)
                ops.hl16_pack_pow2(dZ, dz16, amz, 11)
                ops.hl16_pack_pow2(wflip, wf16, ly['amw'], 14)
                ops.pow2_oscale(osc, amz, 11, ly['amw'], 14)
                ops.conv3x3_raw_hl16(dz16, wf16, zero, dA, L, H, W, cout, cin, osc)
            else:
                ops.conv3x3_raw(dZ, wflip, zero, dA, L, H, W, cout, cin, False)
    return g


class _AppearanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, crops, eng, model, plan, keys, *params):
        P = {k: p.detach().to(torch.float32).contiguous() for k, p in zip(keys, params)}
        c = crops.detach().contiguous()
        feats, tape = appearance_forward_train(eng, model, plan, c, P)
        eng._last_vgg_tape = tape
        ctx.eng, ctx.model, ctx.plan, ctx.tape, ctx.keys, ctx.crops = eng, model, plan, tape, keys, c
        ctx.shapes = [tuple(p.shape) for p in params]
        return feats

    @staticmethod
    def backward(ctx, dF):
        g = appearance_backward(ctx.eng, ctx.model, ctx.plan, ctx.crops, ctx.tape, dF.contiguous())
        out = []
        for k, s in zip(ctx.keys, ctx.shapes):
            out.append(g[k].reshape(s) if k in g else None)
        return (None, None, None, None, None) + tuple(out)


def appearance_autograd(model, plan, crops, update_running_stats=True):
    """Differentiable training-mode ``model.appearance``: crops [L][3][S][S] -> features [L][512]; with
    ``update_running_stats`` every BatchNorm2d buffer takes PyTorch's training-mode momentum update."""
    eng = model.engine()
    keys, ts = _param_list(model)
    eng._last_vgg_tape = None
    out = _AppearanceFn.apply(crops, eng, model, plan, tuple(keys), *ts)
    tape, eng._last_vgg_tape = eng._last_vgg_tape, None
    if update_running_stats and tape is not None:
        with torch.no_grad():
            for ly in tape['layers']:
                bn = ly['bn']
                mean, var = det_batch_stats(ly['L'], ly['rows'])
                m = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(m * mean)
                bn.running_var.mul_(1 - m).add_(m * var)
                bn.num_batches_tracked += 1
    return out
