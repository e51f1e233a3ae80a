"""Drop-in ``nn.Module`` mirror of the reference network API.

Same constructor keywords, same ``state_dict`` keys / shapes, same
``forward(dets, det_info, dets_split)`` signature and return tuple as reference
modules/tracking_net.py:17-35,165-193 - so ``tracking_model.py`` /
``eval_seq.py`` keep working when ``build_model`` imports ``TrackingNet`` from
here.  The torch sub-modules below are *parameter containers only* (they give
the reference key names and ``load_state_dict`` semantics); none of their
``forward`` methods is ever called.  All arithmetic happens in
libmmmot_hip.so through ``Engine``; CPU tensors are rejected (no fallback).

Inference (eval mode) only: the north-star path is the forward.  ``train()`` /
``eval()`` only record the flag like any ``nn.Module`` (the reference's
``TrackingModule`` toggles them around validation, tracking_model.py:32-48,
eval_seq.py:127); a forward in training mode raises - batch-statistics
BatchNorm, dropout and the backward are not built.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import torch_ops
from .engine import Engine
from .pack import VGG_STAGES, pack_weights
from .plan import BatchPlan


def _gn(groups, ch):
    return nn.GroupNorm(groups, ch)


class SkipPool(nn.Module):
    """Parameter layout of reference modules/appear_net.py:9-32 (``fc`` indices 0,1,2,4,5)."""

    def __init__(self, channels, reduction, out_channels, dropblock_size=0):
        super().__init__()
        mid = max(channels // reduction, 64)
        self.channels = channels
        self.fc = nn.Sequential(_gn(1, channels), nn.Conv2d(channels, mid, 1), _gn(1, mid), nn.ReLU(inplace=True),
                                nn.Conv2d(mid, out_channels, 1), _gn(1, out_channels), nn.ReLU(inplace=True))


class AppearanceNet(nn.Module):
    """VGG16-BN + SkipPool image-crop encoder (reference modules/appear_net.py:35-59,130-190)."""

    def __init__(self, arch='vgg', out_channels=512, skippool=True, fpn=False, dropblock=0):
        super().__init__()
        if arch != 'vgg' or not skippool or fpn or out_channels != 512:
            raise NotImplementedError('HIP path builds the vgg16_bn_512 + skippool encoder only '
                                      '(every shipped config; resnet/fpn branches need torchvision downloads)')
        self.arch, self.skippool, self.fpn, self.out_channels, self.dropblock = arch, skippool, fpn, out_channels, dropblock
        stages, heads = [], []
        for stage in VGG_STAGES:
            mods = []
            for (_, cin, cout, pool) in stage:
                mods += [nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
                if pool:
                    mods.append(nn.MaxPool2d(2, 2))
            stages.append(nn.Sequential(*mods))
            heads.append(SkipPool(stage[-1][2], 4, out_channels // 4, 0))
        self.layers = nn.ModuleList(stages)
        self.global_pool = nn.ModuleList(heads)
        self._engine = None

    def forward(self, x):
        """x: L x 3 x S x S crops -> L x 512 (HIP)."""
        eng = _standalone_engine(self, 'appearance.')
        L, S = x.shape[0], x.shape[-1]
        plan = BatchPlan(_dummy_samples(L), S, x.device, use_points=False)
        return torch.ops.mmmot.appearance(x.contiguous(), torch_ops.engine_handle(eng), torch_ops.plan_handle(plan))


class STN3d(nn.Module):
    """Parameter layout of reference modules/point_net.py:47-70."""

    def __init__(self, in_channels, out_size=3, feature_channels=512):
        super().__init__()
        self.out_size = out_size
        self.conv1 = nn.Conv1d(in_channels, 64, 1)
        self.bn1 = _gn(64, 64)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.bn2 = _gn(128, 128)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn3 = _gn(1024, 1024)
        self.idt = nn.Parameter(torch.eye(out_size), requires_grad=False)
        self.fc1 = nn.Linear(1024, 512)
        self.fc_bn1 = _gn(512, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc_bn2 = _gn(256, 256)
        self.output = nn.Linear(256, out_size * out_size)
        nn.init.zeros_(self.output.weight)
        nn.init.zeros_(self.output.bias)


class PointNetfeatGN(nn.Module):
    """Parameter layout of reference modules/point_net.py:89-113."""

    def __init__(self, in_channels=3, out_channels=512, global_feat=True):
        super().__init__()
        self.stn1 = STN3d(in_channels, in_channels, out_channels)
        self.conv1 = nn.Conv1d(in_channels, 64, 1)
        self.bn1 = _gn(64, 64)
        self.conv2 = nn.Conv1d(64, 64, 1)
        self.bn2 = _gn(64, 64)
        self.stn2 = STN3d(64, 64, out_channels)
        self.conv3 = nn.Conv1d(64, 64, 1)
        self.bn3 = _gn(64, 64)
        self.conv4 = nn.Conv1d(64, 128, 1)
        self.bn4 = _gn(128, 128)
        self.conv5 = nn.Conv1d(128, 1024, 1)
        self.bn5 = _gn(1024, 1024)


class PointNet_v1(nn.Module):
    """LiDAR encoder (reference modules/point_net.py:5-44)."""

    def __init__(self, in_channels, out_channels=512, use_dropout=False):
        super().__init__()
        if in_channels not in (3, 4) or out_channels != 512:
            raise NotImplementedError('HIP path builds PointNet_v1(3 | 4 -> 512) (point_len=512)')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.feat = PointNetfeatGN(in_channels, out_channels)
        self.conv1 = nn.Conv1d(1088, 512, 1)
        self.conv2 = nn.Conv1d(512, out_channels, 1)
        self.bn1 = _gn(512, 512)
        self.bn2 = _gn(16, out_channels)
        self.avg_bn = _gn(512, 512)  # unused by the reference forward too; kept for key parity
        self.dropout = None  # eval-only path; reference dropout is identity in eval (point_net.py:29-30)

    def forward(self, x, point_split):
        """x: 1 x 3 x P, point_split: (L+1,) -> (L x 512, [1x3x3, 1x64x64]) (HIP)."""
        eng = _standalone_engine(self, 'point_net.')
        ps = point_split.detach().cpu().numpy().astype(np.int64)
        L = ps.shape[0] - 1
        plan = BatchPlan([(_dummy_samples(L)[0][0], ps)], 32, x.device)
        feat = torch.ops.mmmot.pointnet(x[0].t().contiguous(), torch_ops.engine_handle(eng), torch_ops.plan_handle(plan))
        pn = eng.P['pointnet']
        return feat, [pn['trans1'].unsqueeze(0).clone(), pn['trans2'].unsqueeze(0).clone()]


def _conv_gn(cin, cout, groups):
    return nn.Sequential(nn.Conv1d(cin, cout, 1, 1), _gn(groups, cout))


class _FusionBase(nn.Module):
    mode = None

    def forward(self, objs):
        """objs: 1 x 2D x L -> 3 x D x L (HIP)."""
        eng = _standalone_engine(self, 'fusion_module.', fusion=self.mode)
        L = objs.shape[-1]
        plan = BatchPlan(_dummy_samples(L), 32, objs.device, use_points=False)
        eng.dev = objs.device
        cat = eng.buf('cat', L, 1024)
        cat.copy_(objs[0].t())
        F = eng.buf('F', 3, L, 512)
        eng.fuse(plan, cat, F)
        return F.permute(0, 2, 1).contiguous()


class fusion_module_A(_FusionBase):
    """reference modules/fusion_net.py:73-92"""
    mode = 'A'

    def __init__(self, appear_len, point_len, out_channels):
        super().__init__()
        self.appear_len, self.point_len = appear_len, point_len
        self.input_w = _conv_gn(out_channels * 2, out_channels, out_channels)


class fusion_module_B(_FusionBase):
    """reference modules/fusion_net.py:45-70"""
    mode = 'B'

    def __init__(self, appear_len, point_len, out_channels):
        super().__init__()
        self.appear_len, self.point_len = appear_len, point_len
        self.input_p = _conv_gn(out_channels, out_channels, out_channels)
        self.input_i = _conv_gn(out_channels, out_channels, out_channels)


class fusion_module_C(_FusionBase):
    """reference modules/fusion_net.py:6-42"""
    mode = 'C'

    def __init__(self, appear_len, point_len, out_channels):
        super().__init__()
        self.appear_len, self.point_len = appear_len, point_len
        self.gate_p = nn.Sequential(nn.Conv1d(point_len, point_len, 1, 1), nn.Sigmoid())
        self.gate_i = nn.Sequential(nn.Conv1d(appear_len, appear_len, 1, 1), nn.Sigmoid())
        self.input_p = _conv_gn(point_len, out_channels, out_channels)
        self.input_i = _conv_gn(appear_len, out_channels, out_channels)


FUSION_CLASSES = {'A': fusion_module_A, 'B': fusion_module_B, 'C': fusion_module_C}


class NewEndIndicator_v2(nn.Module):
    """Parameter layout of reference modules/new_end.py:43-60."""

    def __init__(self, in_channels, kernel_size=5, reduction=4, mode='avg'):
        super().__init__()
        self.mode = mode
        mid = min(in_channels, 512)
        self.conv0 = nn.Sequential(nn.Conv2d(in_channels, in_channels, 1, 1), _gn(1, in_channels), nn.ReLU(inplace=True))
        self.conv1 = nn.Sequential(nn.Conv1d(in_channels, mid, 1, 1), _gn(1, mid), nn.ReLU(inplace=True),
                                   nn.Conv1d(mid, in_channels // reduction, 1, 1), _gn(1, in_channels // reduction),
                                   nn.ReLU(inplace=True), nn.Conv1d(in_channels // reduction, 1, 1, 1), nn.Sigmoid())


class affinity_module(nn.Module):
    """Pairwise affinity + new/end heads (reference modules/gcn.py:45-82)."""

    def __init__(self, in_channels, new_end=None, affinity_op='multiply'):
        super().__init__()
        if in_channels != 512:
            raise NotImplementedError('HIP path builds the 512-channel affinity module')
        self.in_channels = in_channels
        self.affinity_op = affinity_op
        self.w_new_end = new_end(in_channels) if new_end is not None else NewEndIndicator_v2(in_channels)
        c = in_channels
        self.conv1 = nn.Sequential(nn.Conv2d(c, c, 1, 1), _gn(c, c), nn.ReLU(inplace=True),
                                   nn.Conv2d(c, c, 1, 1), _gn(c, c), nn.ReLU(inplace=True),
                                   nn.Conv2d(c, c // 4, 1, 1), _gn(c // 4, c // 4), nn.ReLU(inplace=True),
                                   nn.Conv2d(c // 4, 1, 1, 1))

    def forward(self, objs, dets):
        """objs R x D x N, dets R x D x M -> (R x 1 x N x M logits, R x M new, R x N end) (HIP)."""
        eng = _standalone_engine(self, 'w_link.', affinity_op=self.affinity_op)
        R, _, N = objs.shape
        M = dets.shape[-1]
        plan = BatchPlan([([N, M], None)], 32, objs.device, rows=tuple(range(R)), use_points=False)
        eng.dev = objs.device
        F = eng.buf('F', R, N + M, 512)
        F[:, :N].copy_(objs.permute(0, 2, 1))
        F[:, N:].copy_(dets.permute(0, 2, 1))
        link, new, end = eng.affinity(plan, F)
        return link.view(R, 1, N, M), new[:, N:].clone(), end[:, :N].clone()


def _dummy_samples(L):
    """One sample of L detections split over two frames (sub-module calls carry no frame split)."""
    if L < 2:
        raise ValueError('stand-alone sub-module calls need >= 2 detections')
    return [([L // 2, L - L // 2], None)]


def _standalone_engine(module, prefix, **cfg):
    """Engine for calling a sub-module on its own (module-level parity tests)."""
    ver = tuple(p._version for p in module.state_dict().values())
    cache = getattr(module, '_eng_cache', None)
    if cache is None or cache[0] != ver:
        from .ops import HipOps
        dev = next(module.parameters()).device
        sd = {prefix + k: v for k, v in module.state_dict().items()}
        packed = pack_weights(sd, cfg.get('fusion', 'A'), dev)
        eng = Engine(packed, HipOps(), **cfg)
        object.__setattr__(module, '_eng_cache', (ver, eng))
        cache = module._eng_cache
    return cache[1]


def _collect_tensors(obj, out=None):
    """every tensor reachable through dicts / lists / tuples of a packed-weight tree"""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _collect_tensors(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _collect_tensors(v, out)
    return out


def _copy_packed_(dst, src):
    """Copy the packed tree ``src`` into ``dst`` in place (same keys / shapes / dtypes); False when they differ."""
    if torch.is_tensor(dst):
        if not torch.is_tensor(src) or dst.shape != src.shape or dst.dtype != src.dtype or dst.device != src.device:
            return False
        dst.copy_(src)
        return True
    if isinstance(dst, dict):
        if not isinstance(src, dict) or dst.keys() != src.keys():
            return False
        ok = True
        for k in dst:
            if torch.is_tensor(dst[k]) or isinstance(dst[k], (dict, list, tuple)):
                ok = _copy_packed_(dst[k], src[k]) and ok
            elif dst[k] != src[k]:
                dst[k] = src[k]  # python scalars (scales): the kernels take them by value at launch time ...
                ok = False       # ... so a captured launch keeps the old one
        return ok
    if isinstance(dst, (list, tuple)):
        if not isinstance(src, (list, tuple)) or len(dst) != len(src):
            return False
        return all([_copy_packed_(d, s_) for d, s_ in zip(dst, src)])
    return dst == src


class GraphedForward:
    """See ``TrackingNet.capture``."""

    def __init__(self, model, plan, crops, points):
        self.plan = plan
        self.crops = None if crops is None else crops.clone()
        self.points = None if points is None else points.clone()
        eng = model.engine()
        with torch.no_grad():
            model.forward_batch(plan, self.crops, self.points)   # warm-up: workspace allocation, range-guard check
            torch.cuda.synchronize()
            # (the range guard sees the capture and only binds the engine's counter block: check_range() reads it)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.result = model.forward_batch(plan, self.crops, self.points)
        # the graph holds raw pointers into the engine's packed weights and workspace arena: keep BOTH alive (every
        # tensor of eng.P and eng.ws at capture time) even if the model is re-packed or a later, larger forward
        # re-allocates workspace buffers - a replay then reads valid memory.  It would still compute with the weights
        # of capture time, so a re-pack that could not be done in place (invalidate / load_state_dict / set_trunk / a
        # head refresh that changed shapes) marks the graph stale: __call__ raises instead of silently replaying it.
        # refresh_head() itself copies into the packed tensors IN PLACE, so a graph follows an optimizer step.
        self._engine = eng
        self._model = model
        self._pack_version = model._pack_version
        self._keep = list(eng.ws.values()) + _collect_tensors(eng.P) + [eng._range_buf]

    def stale(self):
        m = self._model
        return m._engine is not self._engine or m._pack_version != self._pack_version

    def check_range(self):
        """The trunk's range guard never runs inside a replay (its counter read synchronises): call this now and then
        when serving from a graph.  Returns the counters (e4m3-saturated, fp16-clamped, conv1_1 hits) accumulated by
        this engine since the last check and raises if the captured arithmetic left its range (re-capture after
        ``model.set_trunk('f32')`` or with rescaled inputs)."""
        eng = self._engine
        sat, clamp, c11 = eng.read_range(reset=True)
        if clamp > 0 or (c11 > 0 and eng.trunk != 'f16q8'):
            raise RuntimeError('mmmot_amd: %d activation elements left the fp16 range inside a captured forward (trunk %s): '
                               'the replayed results are wrong; lower the trunk (set_trunk) and capture again'
                               % (clamp + c11, eng.trunk))
        return sat, clamp, c11

    def __call__(self, crops=None, points=None):
        if self.stale():
            raise RuntimeError('mmmot_amd: this captured forward is stale - the model was re-packed (load_state_dict / '
                               'invalidate / set_trunk / .to()) after capture(); capture again')
        if crops is not None and crops.data_ptr() != self.crops.data_ptr():
            self.crops.copy_(crops, non_blocking=True)
        if points is not None and points.data_ptr() != self.points.data_ptr():
            self.points.copy_(points, non_blocking=True)
        self.graph.replay()
        return self.result


class TrackingNet(nn.Module):
    """Reference-compatible tracking network (modules/tracking_net.py:15-193), HIP forward."""

    def __init__(self, seq_len, appear_len=512, appear_skippool=False, appear_fpn=False, score_arch='vgg',
                 score_fusion_arch='C', appear_arch='vgg', point_arch='v1', point_len=512,
                 softmax_mode='single', test_mode=0, affinity_op='multiply', dropblock=5, end_arch='v2',
                 end_mode='avg', without_reflectivity=True, neg_threshold=0, use_dropout=False):
        super().__init__()
        if appear_len != 512 or point_len != 512 or point_arch != 'v1' or end_arch != 'v2':
            raise NotImplementedError('HIP path covers appear_len=point_len=512, point_arch=v1, end_arch=v2 '
                                      '(all shipped configs); single-modality runs use forward_rows()')
        if score_arch not in ('branch_cls', 'branch_reg'):
            raise NotImplementedError("score_arch must be 'branch_cls' or 'branch_reg' (reference builds no w_det otherwise)")
        if score_fusion_arch not in FUSION_CLASSES:
            raise ValueError('unknown score_fusion_arch %r' % (score_fusion_arch,))
        self.seq_len = seq_len
        self.score_arch = score_arch
        self.neg_threshold = neg_threshold
        self.test_mode = test_mode  # read by the host tracker (tracking_model.py:19-22)
        self.softmax_mode = softmax_mode
        self.affinity_op = affinity_op
        self.score_fusion_arch = score_fusion_arch
        self.end_mode = end_mode
        self.fusion_module = FUSION_CLASSES[score_fusion_arch](appear_len, point_len, out_channels=point_len)
        self.appearance = AppearanceNet(appear_arch, appear_len, skippool=appear_skippool, fpn=appear_fpn,
                                        dropblock=dropblock)
        self.point_net = PointNet_v1(4 - int(without_reflectivity), out_channels=point_len, use_dropout=use_dropout)
        self.w_link = affinity_module(
            point_len, new_end=lambda c: NewEndIndicator_v2(c, kernel_size=5, reduction=4, mode=end_mode),
            affinity_op=affinity_op)
        c = point_len
        self.w_det = nn.Sequential(nn.Conv1d(c, c, 1, 1), nn.BatchNorm1d(c), nn.ReLU(inplace=True),
                                   nn.Conv1d(c, c // 2, 1, 1), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True),
                                   nn.Conv1d(c // 2, 1, 1, 1))
        self.trunk = os.environ.get('MMMOT_TRUNK', 'f16x3')
        self._ops = None       # operator backend (HipOps unless a test injects another)
        self._engine = None
        self._engine_key = None
        self._plans = {}
        self._img_plans = {}   # image-only plans of forward()'s trunk-first launch, keyed by (frame counts, crop side)
        # forward(): launch the trunk before the point split is read back and the plan is built (MMMOT_IMAGE_FIRST=0: after)
        self.image_first = os.environ.get('MMMOT_IMAGE_FIRST', '1') != '0'
        self.freeze_appearance = False  # training mode: True = frozen eval-mode image features instead of the training trunk
        self._pack_version = 0     # bumped whenever packed weights are REPLACED (captured graphs go stale)
        self._head_versions = None  # parameter versions the live engine's head was packed from

    # ---- backend / packing ---------------------------------------------------
    def set_ops(self, ops):
        """Inject an operator backend (tests use a torch emulation of the C-ABI to
        check the host logic on CPU; the product default is HipOps)."""
        self._ops = ops
        self._engine = None

    def engine(self):
        """Packed-weight engine; rebuilt after load_state_dict / .to() / .cuda() (call
        ``invalidate()`` after editing parameters in place)."""
        if self._engine is None:
            dev = next(self.parameters()).device
            if self._ops is None:
                from .ops import HipOps
                self._ops = HipOps()  # raises if libmmmot_hip.so is missing: no fallback
            packed = pack_weights(self.state_dict(), self.score_fusion_arch, dev)
            self._engine = Engine(packed, self._ops, fusion=self.score_fusion_arch, affinity_op=self.affinity_op,
                                  softmax_mode=self.softmax_mode, neg_threshold=self.neg_threshold,
                                  score_arch=self.score_arch, end_mode=self.end_mode, trunk=self.trunk)
            self._plans = {}
            self._pack_version += 1
            self._head_versions = self._current_head_versions()
            self._snapshot_versions()
        return self._engine

    def _snapshot_versions(self):
        """Versions of EVERY parameter and buffer the live engine was packed from (the head's versions alone
        miss encoder-only training and the 'eval, one training step, eval' flow).  The tensor list is kept so that the
        per-forward check is one list comprehension (25 us), not a module walk (0.4 ms)."""
        heads = set()
        for name in self._HEADS:
            m = getattr(self, name)
            heads.update(id(t) for t in list(m.parameters()) + list(m.buffers()))
        self._packed_tensors = list(self.parameters()) + list(self.buffers())
        self._packed_is_head = [id(t) in heads for t in self._packed_tensors]
        self._packed_versions = [t._version for t in self._packed_tensors]

    def _packed_is_current(self):
        return self._engine is None or [t._version for t in self._packed_tensors] == self._packed_versions

    def _repack_if_trained(self):
        """Eval-mode entry points: everything the inference engine holds (folded BatchNorm, fp16-split copies, folded
        transforms) is stale once any parameter / buffer was modified in place since it was packed - by an optimizer step,
        a training-mode BatchNorm update, or by hand."""
        if getattr(self, '_trained_since_pack', False) or not self._packed_is_current():
            self._trained_since_pack = False
            self.invalidate()

    _HEADS = ('fusion_module', 'w_det', 'w_link')

    def _current_head_versions(self):
        # parameters and buffers of the three head modules, in module order (a state_dict() walk costs 15 ms)
        out = []
        for name in self._HEADS:
            m = getattr(self, name)
            out += [p._version for p in m.parameters()] + [b._version for b in m.buffers()]
        return tuple(out)

    def refresh_head_device(self):
        """What a TRAINING step needs after ``optimizer.step()``: the fp32 packed tensors of fusion_module and w_link
        recomputed from the live parameters ON THE DEVICE (concatenations and reshapes - no host round trip except the
        two scalar output biases) and copied into the engine's tensors in place.  refresh_head() does the same on the
        host in float64 and also rebuilds the fp16-split copies (0.2 s per call: 95 % of a training step, measured);
        here those copies go stale, so the training forward runs its GEMMs on the fp32 weights (Engine.fp32_mlp) and the
        next EVAL forward re-packs everything (``_trained_since_pack``)."""
        eng = self.engine()
        with torch.no_grad():
            fm, lk = self.fusion_module, self.w_link
            fl = lambda conv: conv.weight.detach().flatten(1)
            fu = eng.P['fusion']
            new = {}
            if self.score_fusion_arch == 'A':
                new.update(w0=fl(fm.input_w[0]), b0=fm.input_w[0].bias, g0=fm.input_w[1].weight, be0=fm.input_w[1].bias)
            elif self.score_fusion_arch == 'B':
                for j, m in enumerate((fm.input_p, fm.input_i)):
                    new.update({'w%d' % j: fl(m[0]), 'b%d' % j: m[0].bias, 'g%d' % j: m[1].weight, 'be%d' % j: m[1].bias})
            else:
                for j, (gt, inp) in enumerate(((fm.gate_p, fm.input_p), (fm.gate_i, fm.input_i))):
                    one, zero = torch.ones_like(inp[1].weight), torch.zeros_like(inp[1].bias)
                    new.update({'w%d' % j: torch.cat([fl(gt[0]), fl(inp[0])], 0), 'b%d' % j: torch.cat([gt[0].bias, inp[0].bias], 0),
                                'g%d' % j: torch.cat([one, inp[1].weight], 0), 'be%d' % j: torch.cat([zero, inp[1].bias], 0)})
            for k, v in new.items():
                fu[k].copy_(v.detach())
            ne, c1, P = lk.w_new_end, lk.conv1, eng.P['w_link']
            pairs = dict(wa=torch.cat([fl(ne.conv0[0]), fl(c1[0])], 0), ba=torch.cat([ne.conv0[0].bias, c1[0].bias], 0),
                         g_ne0=ne.conv0[1].weight, be_ne0=ne.conv0[1].bias, g1=c1[1].weight, be1=c1[1].bias,
                         w3=fl(c1[3]), b3=c1[3].bias, g4=c1[4].weight, be4=c1[4].bias, w6=fl(c1[6]), b6=c1[6].bias,
                         g7=c1[7].weight, be7=c1[7].bias, w9=c1[9].weight.reshape(-1),
                         nw0=fl(ne.conv1[0]), nb0=ne.conv1[0].bias, ng1=ne.conv1[1].weight, nbe1=ne.conv1[1].bias,
                         nw3=fl(ne.conv1[3]), nb3=ne.conv1[3].bias, ng4=ne.conv1[4].weight, nbe4=ne.conv1[4].bias,
                         nw6=ne.conv1[6].weight.reshape(-1))
            for k, v in pairs.items():
                P[k].copy_(v.detach())
            P['b9'], P['nb6'] = float(c1[9].bias.item()), float(ne.conv1[6].bias.item())
        eng._h16_head_stale = True
        self._trained_since_pack = True
        self._pack_version += 1  # a captured eval graph would replay the stale fp16-split copies
        self._head_versions = self._current_head_versions()
        return eng

    def head_is_current(self):
        """False once a head parameter / buffer was modified in place (``optimizer.step()``, the BatchNorm momentum
        update) after the engine packed it."""
        return self._engine is not None and self._head_versions == self._current_head_versions()

    def set_trunk(self, trunk):
        """Arithmetic of the VGG trunk (the other GEMMs follow: f16x3, or exact fp32 with 'f32'):
        'f16x3' (default): fp16 matrix cores, 3-term hi/lo split - fp32-class, score error ~3e-5 against the fp32
                 reference (indistinguishable from fp32 summation-order noise);
        'f16q8': opt-in: fp16 main term + both correction terms in one block-scaled fp8 MFMA on the trunk layers
                 ``Engine.q8_layers`` (+20 % throughput; score error 3e-4 on He-normal weights, up to 9e-4 on
                 trained-like statistics with every layer in fp8; crops under 64 pixels run f16x3);
        'f32':   exact fp32 MFMA everywhere.
        The engine's range guard lowers f16q8 -> f16x3 -> f32 by itself when activations leave the e4m3 / fp16 range
        (``engine().range_events``)."""
        self.trunk = trunk
        self.invalidate()

    def invalidate(self):
        self._engine = None
        self._plans = {}
        self.__dict__.pop('_train_plans', None)  # training-mode plans (mmmot_amd/train.py::forward_train)
        self._pack_version += 1

    def refresh_head(self):
        """Re-pack only the head's weights (fusion_module, w_det, w_link: 3 M of the 21 M parameters) into the live
        engine - what a training step on the head needs after ``optimizer.step()`` (mmmot_amd/backward.py); the
        encoders' packed weights (VGG hl16 / hq8 copies, PointNet) stay as they are."""
        eng = self.engine()
        sd = {k: v for k, v in self.state_dict().items() if k.split('.')[0] in self._HEADS}
        P = pack_weights(sd, self.score_fusion_arch, next(self.parameters()).device)
        # in place where the packed layout is unchanged (always, for a pure weight update): the packed tensors keep
        # their addresses, so a captured hipGraph (GraphedForward) computes with the new weights on its next replay;
        # anything that cannot be copied in place replaces the entry and marks captured graphs stale
        inplace = True
        for k in ('fusion', 'w_det', 'w_link'):
            if not _copy_packed_(eng.P[k], P[k]):
                eng.P[k] = P[k]
                inplace = False
        if not inplace:
            self._pack_version += 1
        self._head_versions = self._current_head_versions()
        for i, t in enumerate(self._packed_tensors):  # the head is current again; the encoders keep their snapshot
            if self._packed_is_head[i]:
                self._packed_versions[i] = t._version
        return eng

    def _apply(self, fn, *a, **k):
        self.invalidate()  # .to() / .cuda(): packed weights and the cached plans' tables live on the old device
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **k):
        self._engine = None
        self._pack_version += 1
        return super().load_state_dict(state_dict, strict=strict, **k)

    def make_plan(self, samples, crop_hw, rows=(0, 1, 2)):
        """samples: list of (frame_counts, points_split) - see BatchPlan."""
        dev = next(self.parameters()).device
        return BatchPlan(samples, crop_hw, dev, rows=rows, use_points=(1 in rows or 2 in rows))

    # ---- batched entry (no reference counterpart: the reference batch is 1) --
    def forward_batch(self, plan, crops, points):
        """crops [Lt,3,S,S], points [P,3] (device, concatenated over the plan's samples).
        Returns per-sample reference-shaped tuples."""
        self._repack_if_trained()
        if self.training:
            raise NotImplementedError('forward_batch / forward_rows compute the eval-mode forward (call .eval()); the '
                                      'training-mode forward of tracking_model.py:50-66 is model(dets, det_info, dets_split) '
                                      'after .train() - mmmot_amd/train.py')
        eng = self.engine()
        if eng.ops.name == 'hip':
            # the registered PyTorch-ROCm operator (mmmot_amd/torch_ops.py): CUDA dispatch key only, no CPU kernel
            det_t, link_t, new_t, end_t = torch.ops.mmmot.forward_batch(
                crops, points, torch_ops.engine_handle(eng), torch_ops.plan_handle(plan))
            out = dict(det=det_t, link=link_t, new=new_t, end=end_t)
        else:  # an injected backend (tests: the torch emulation of the C-ABI)
            out = eng.forward(plan, crops, points)
        res = []
        pi = 0
        nR = plan.nR
        for b, fc in enumerate(plan.frame_counts):
            d0, d1 = int(plan.det_off[b]), int(plan.det_off[b + 1])
            links = []
            for f in range(len(fc) - 1):
                _, _, N, _, M = plan.pairs[pi]
                o = plan.link_off[pi]
                links.append(out['link'][o:o + nR * N * M].view(nR, N, M))
                pi += 1
            res.append((out['det'][:, d0:d1], links, out['new'][:, d0:d1], out['end'][:, d0:d1]))
        return res

    def capture(self, plan, crops, points):
        """hipGraph of one ``forward_batch`` for a FIXED plan (the C-ABI entry points only launch: no allocation, no
        synchronisation, so the whole step is capturable).  Returns a ``GraphedForward``; its ``__call__(crops,
        points)`` copies the inputs into the captured buffers, replays the graph and returns the same per-sample
        tuples as ``forward_batch`` (views of static output buffers, overwritten by the next replay).  Use it when
        consecutive calls share the batch shape (throughput serving of fixed-size batches, the benchmark); a tracker
        whose detections change every frame calls ``forward`` / ``forward_batch`` instead."""
        return GraphedForward(self, plan, crops, points)

    def trans(self):
        pn = self.engine().P['pointnet']
        return [pn['trans1'].unsqueeze(0), pn['trans2'].unsqueeze(0)]

    # ---- reference entry ------------------------------------------------------
    def forward(self, dets, det_info, dets_split):
        """Same contract as reference modules/tracking_net.py:165-193: in eval mode
        (det_scores 3xL, [link_scores 3xNxM ...], new_scores 3xL, end_scores 3xL, trans); in training mode
        (``.train()``) the differentiable forward of mmmot_amd/train.py - raw det scores, new / end scores without the
        eval padding, batch-statistics BatchNorm in the trunk and w_det (``self.freeze_appearance = True`` runs the image
        branch frozen in eval mode instead)."""
        if self.training:
            from .train import forward_train
            return forward_train(self, dets, det_info, dets_split)
        return self.forward_rows(dets, det_info, dets_split, rows=(0, 1, 2))

    def _trunk_first(self, fc, S, crops):
        """The point split is on the HOST already (a pipeline that prepared the points itself, mmmot_amd/pipeline.py): no
        read-back to hide, but the image branch still launches first so that PointNet's small launches run on the engine's
        side stream beside the trunk instead of behind it.  Returns False when the engine is not the HIP one."""
        self._repack_if_trained()
        eng = self.engine()
        if eng.ops.name != 'hip':
            return False
        key = (tuple(fc), S, str(crops.device))
        plan_img = self._img_plans.get(key)
        if plan_img is None:
            if len(self._img_plans) > 256:
                self._img_plans.clear()
            plan_img = BatchPlan([(fc, None)], S, crops.device, rows=(0,), use_points=False)
            self._img_plans[key] = plan_img
        eng.image_first(plan_img, crops)
        return True

    def _split_behind_trunk(self, ps_t, fc, S, crops):
        """The point split (device tensor) is needed on the host to build the plan; the trunk is not waiting for it.  Copy it
        on a side stream, launch the image branch (Engine.image_first: its tables depend on the frame counts only and are
        cached per (counts, crop side)), then wait for the copy alone: the read-back and the plan build that follows run
        beside ~2 ms of trunk time instead of in front of it.  Returns the split as int64 numpy, or None when the engine is
        not the HIP one (injected test backends take the plain path)."""
        self._repack_if_trained()
        eng = self.engine()
        if eng.ops.name != 'hip':
            return None
        dev = crops.device
        n = int(ps_t.numel())
        st = getattr(self, '_split_stage', None)
        if st is None or st[0].numel() < n or st[0].dtype != ps_t.dtype or st[1].device != dev:
            st = (torch.empty(max(64, 2 * n), dtype=ps_t.dtype).pin_memory(), torch.cuda.Stream(dev), torch.cuda.Event())
            self._split_stage = st
        host, side, ev = st
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)  # the producer of points_split is ordered before the copy; nothing of this forward is queued yet
        with torch.cuda.stream(side):
            host[:n].copy_(ps_t.detach(), non_blocking=True)
            ev.record(side)
        key = (tuple(fc), S, str(dev))
        plan_img = self._img_plans.get(key)
        if plan_img is None:
            if len(self._img_plans) > 256:
                self._img_plans.clear()
            plan_img = BatchPlan([(fc, None)], S, dev, rows=(0,), use_points=False)
            self._img_plans[key] = plan_img
        eng.image_first(plan_img, crops)
        ev.synchronize()
        return host[:n].numpy().astype(np.int64)

    def forward_rows(self, dets, det_info, dets_split, rows=(0, 1, 2)):
        """Single-modality variant: rows=(0,) image-only skips PointNet+fusion, rows=(1,) LiDAR-only
        skips VGG+fusion; the returned tensors hold only the requested modality rows."""
        fc = [int(d.item()) if torch.is_tensor(d) else int(d) for d in dets_split]
        rows = tuple(rows)
        need_pts = (1 in rows) or (2 in rows)
        need_img = (0 in rows) or (2 in rows)
        # dets: the reference's normalised fp32 [L,3,S,S] crops, or the uint8 [L,S,S,3] crops of the resize
        # (mmmot_amd.crops.crop_resize_u8): ToTensor / Normalize then happen inside the first trunk launch
        S = (int(dets.shape[1]) if dets.dtype == torch.uint8 else int(dets.shape[-1])) if dets is not None else 0
        ps = None
        points = None
        crops = dets.contiguous() if need_img else None
        beside = False  # the trunk of this call is already running (Engine.image_first)
        if need_pts:
            # (before the trunk is launched: a layout copy queued behind it would not be seen by the LiDAR branch, which
            # runs on a side stream beside the trunk)
            points = det_info['points']
            points = points.reshape(-1, points.shape[-1]).contiguous()  # [P][3] or [P][4] (with reflectivity)
            ps_t = det_info['points_split'].reshape(-1)
            if (need_img and ps_t.is_cuda and crops.is_cuda and self.image_first and not self.training
                    and not torch.cuda.is_current_stream_capturing()):
                ps = self._split_behind_trunk(ps_t, fc, S, crops)
                beside = ps is not None
            elif (need_img and not ps_t.is_cuda and crops.is_cuda and points.is_cuda and self.image_first
                  and not self.training and not torch.cuda.is_current_stream_capturing()):
                beside = self._trunk_first(fc, S, crops)
            if ps is None:
                ps = ps_t.detach().to('cpu').numpy().astype(np.int64)  # one D2H copy (reference: 2 .item() per detection)
        try:
            return self._forward_rows_tail(fc, ps, S, rows, need_pts, crops, points, dets, beside)
        except BaseException:
            if beside and self._engine is not None:  # the image branch that was issued belongs to no forward any more
                self._engine._image_token = None
                self._engine._pre_image = None
            raise

    def _forward_rows_tail(self, fc, ps, S, rows, need_pts, crops, points, dets, beside):
        dev = points.device if points is not None else dets.device
        key = (tuple(fc), None if ps is None else ps.tobytes(), S, rows, str(dev))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 64:
                self._plans.clear()
            if beside:
                # the plan's tables are uploaded on the engine's side stream - the stream the LiDAR branch will run on -
                # instead of behind the trunk; the main stream (pairwise head) waits for that upload alone
                side = self.engine()._side_stream(dev)
                with torch.cuda.stream(side):
                    plan = BatchPlan([(fc, ps)], S, dev, rows=rows, use_points=need_pts)
                torch.cuda.current_stream(dev).wait_stream(side)
            else:
                plan = BatchPlan([(fc, ps)], S, dev, rows=rows, use_points=need_pts)
            self._plans[key] = plan
        det, links, new, end = self.forward_batch(plan, crops, points)[0]
        trans = self.trans() if need_pts else None
        return det, links, new, end, trans
