"""Launch schedule of the mmMOT forward over the device operator layer.

One ``Engine`` owns the packed weights of a ``TrackingNet`` and a workspace
arena; ``forward(plan, crops, points)`` issues the kernel sequence for a whole
batch of samples on the current HIP stream and returns device tensors.  The
schedule follows the data flow of reference modules/tracking_net.py:128-193
(feature -> determine_det -> associate) but not its module structure: every
normalisation on the path is a whole-sample reduction, so the schedule is a
chain of  GEMM(+stats epilogue) -> finalize -> next GEMM(normalise prologue).

``ops`` is the operator backend: ``mmmot_amd.ops.HipOps`` in the product.  The
engine itself only allocates memory and orders launches.
"""
import contextlib
import os
import threading
import warnings

import numpy as np
import torch

from .ops import (ACT_NONE, ACT_RELU, ACT_SIGMOID, A_NORM_RELU, A_PAIR, A_PLAIN, FUSION_MODES, PAIR_OPS,
                  SOFTMAX_MODES)

EPS = 1e-5  # nn.GroupNorm / nn.BatchNorm default used everywhere in the reference


class Engine:
    def __init__(self, packed, ops, fusion='A', affinity_op='multiply', softmax_mode='none',
                 neg_threshold=0.0, score_arch='branch_cls', end_mode='avg', trunk=None):
        trunk = trunk or os.environ.get('MMMOT_TRUNK', 'f16x3')
        if trunk not in ('f16x3', 'f16q8', 'f32'):
            raise ValueError("trunk must be 'f16x3' (fp16 matrix cores, 3-term split), 'f16q8' (fp16 main term + "
                             "fp8 correction terms, include/mmmot_hip.h hq8) or 'f32' (exact fp32 MFMA)")
        self.trunk = trunk            # arithmetic the trunk currently runs in (the range guard may lower it)
        self.trunk_requested = trunk
        # Range guard (include/mmmot_hip.h: mmmot_trunk_range_read).  The hq8 / hl16 activation formats have a finite
        # range: e4m3 copies saturate above 1792 (products of that element become fp16-class), the fp16 `hi` half
        # clamps at 65000 (wrong value).  The trunk epilogues count both on the device; the engine reads the counters
        # synchronously on its first forward (and repeats that forward's trunk when they are hit), asynchronously after
        # every later one (a 16-byte read-back inspected at the next forward: no synchronisation, one step of lag),
        # and moves the trunk f16q8 -> f16x3 -> f32 for good.  MMMOT_RANGE_CHECK_EVERY=n adds a synchronous
        # check-and-recompute every n forwards; MMMOT_RANGE_GUARD=0 disables the guard.  See _guarded_appearance.
        self.range_guard = os.environ.get('MMMOT_RANGE_GUARD', '1') != '0'
        self.range_check_every = int(os.environ.get('MMMOT_RANGE_CHECK_EVERY', '0'))
        self.q8_sat_limit = float(os.environ.get('MMMOT_Q8_SAT_LIMIT', '1e-4'))  # tolerated fraction of saturated elements
        self.range_events = []
        self.last_out_of_range_forward = None  # index of the latest forward whose out-of-range results were already returned
        self._n_forward = 0
        self._range_buf = self._range_host = self._range_pending = None
        self._range_win0 = 0  # first forward whose trunk launches no inspected / queued read-back covers yet
        self.out_of_range_window = None  # (first, last) forward indices of the latest asynchronous detection
        self._busy = threading.Lock()  # see forward()
        self._last_stream = None
        self._image_token = None  # image_first(): the image branch of the next forward is already in the workspace
        self._pre_image = None    # ... and the event recorded in front of it (where the LiDAR branch may start from)
        # f16q8 only: trunk layers (indices into P['vgg'], 1..12) that run the hq8 arithmetic; None = all of them
        # (MMMOT_Q8_LAYERS=all).  The others run f16x3; at a boundary the activation tensor is re-encoded (hq8 <-> hl16,
        # two small kernels).  Default: conv3_1 .. conv5_3 (layers 4..12).  Measured on trained-like statistics
        # (tools/study_robustness.py, calibrated BatchNorm, per-channel gains over 1e4, heavy-tailed weights): the
        # full-resolution layers 1..3 carry two thirds of the e4m3 error (all layers 5.8e-4 .. 9.5e-4 of the 1e-3
        # budget; layers 4..12 3.8e-4 .. 4.0e-4) for a fifth of the trunk's time.
        ql = os.environ.get('MMMOT_Q8_LAYERS', '4,5,6,7,8,9,10,11,12')
        self.q8_layers = None if ql == 'all' else set(int(v) for v in ql.split(',') if v)
        # PointNet conv5 / conv1: statistics pass + fused normalise-ReLU-segment-sum pass instead of
        # materialising the [P][1024] / [P][512] tensors (MMMOT_PN_FUSED=0 keeps the materialising path)
        self.pn_fused = os.environ.get('MMMOT_PN_FUSED', '1') != '0'
        # conv5 statistics from the Gram matrix of its input instead of a statistics pass of the GEMM
        self.pn_gram = os.environ.get('MMMOT_PN_GRAM', '1') != '0'
        # conv2..conv4 (K = 64) from the persistent weight-resident kernel (pn_mlp64.hip) instead of the generic row GEMM
        self.pn_mlp64 = os.environ.get('MMMOT_PN_MLP64', '1') != '0'
        # LiDAR branch on a side stream next to the trunk: opt-in (MMMOT_TWO_STREAMS=1).  Measured (profiles/README.md
        # r02): cfg3 x 8 pairs 275.3 -> 276.0 pairs/s (noise) while every trunk launch's own duration grows by the time
        # it waits for CUs; B = 1 hipGraph replay 2.85 -> 2.71 ms, eager latency unchanged.
        self.two_streams = os.environ.get('MMMOT_TWO_STREAMS', '0') == '1'
        # forward() after image_first(): PointNet on a side stream beside the already running trunk (MMMOT_PN_BESIDE_TRUNK=0: behind it)
        self.pn_beside_trunk = os.environ.get('MMMOT_PN_BESIDE_TRUNK', '1') != '0'
        self._side = {}
        # SkipPool heads: one launch per stage (MMMOT_SP_FUSED=0: LayerNorm / GEMM / LayerNorm / GEMM / LayerNorm launches)
        self.sp_fused = os.environ.get('MMMOT_SP_FUSED', '1') != '0'
        # conv1_1 evaluated inside conv1_2's patch prologue (MMMOT_FUSE_CONV1=0: two launches)
        self.fuse_conv1 = os.environ.get('MMMOT_FUSE_CONV1', '1') != '0'
        # f16q8 applies to crops of at least this side; smaller crops run the f16x3 trunk.  The e4m3 correction
        # terms cost ~3e-4 of score error at the 64..224-pixel crops of the reference's configurations and more
        # where less spatial averaging follows the trunk (6e-4 measured at 32-pixel crops, tests/test_hq8_gpu.py),
        # while the trunk is a negligible part of the step there.
        self.q8_min_crop = int(os.environ.get('MMMOT_Q8_MIN_CROP', '64'))
        self.mlp = 'f32' if trunk == 'f32' else 'f16x3'  # the 1x1-conv / linear GEMMs: exact fp32 only with the f32 trunk
        if affinity_op not in PAIR_OPS:
            raise ValueError('unknown affinity_op %r' % (affinity_op,))
        if softmax_mode not in SOFTMAX_MODES and softmax_mode != 'none':
            softmax_mode = 'none'  # reference falls through to the raw logits (tracking_net.py:123-124)
        self.end_mode = end_mode  # 'avg': mean over the prev / curr axis, anything else: maximum (new_end.py:69-74)
        self.P = packed
        self.ops = ops
        self.fusion = fusion
        self.affinity_op = affinity_op
        self.softmax_mode = softmax_mode
        self.neg_threshold = float(neg_threshold)
        self.score_arch = score_arch
        self.ws = {}
        self.dev = None
        self.keep = None         # optional dict collecting per-stage tensors (tests)
        self.conv_events = None  # optional list collecting per-launch HIP events (bench.py)

    # ---- workspace arena ---------------------------------------------------
    def buf(self, name, *shape, device=None):
        n = 1
        for s in shape:
            n *= int(s)
        t = self.ws.get(name)
        if t is None or t.numel() < n or (device is not None and t.device != torch.device(device)):
            t = torch.empty(max(n, 4), dtype=torch.float32, device=device if device is not None else self.dev)
            self.ws[name] = t
        return t[:n].view(*shape)

    def buf64(self, name, *shape):
        n = 1
        for s in shape:
            n *= int(s)
        t = self.ws.get(name)
        if t is None or t.numel() < n or t.dtype != torch.float64:
            t = torch.empty(max(n, 4), dtype=torch.float64, device=self.dev)
            self.ws[name] = t
        return t[:n].view(*shape)

    def _finalize(self, name, part, tiles, C, NG, gamma, beta):
        sc = self.buf(name + '_sc', tiles.G, C)
        sh = self.buf(name + '_sh', tiles.G, C)
        self.ops.gn_finalize(part, tiles, C, NG, gamma, beta, EPS, sc, sh)
        return sc, sh

    def _gemm(self, d, name, tiles, N, K, **kw):
        """Row GEMM with weight d[name]: on the fp16 matrix cores (3-term hi/lo split, hl16 weight copy
        made by pack._add_hl16_copies) when mlp == 'f16x3', else on the exact fp32 MFMA."""
        if self.mlp == 'f16x3' and (name + '_h16') in d:
            self.ops.gemm(d[name + '_h16'], tiles, N, K, w_hl16=True, oscale=d[name + '_os'], **kw)
        else:
            self.ops.gemm(d[name], tiles, N, K, **kw)

    @contextlib.contextmanager
    def fp32_mlp(self):
        """row GEMMs on the fp32 weights inside the block (the training forward: the fp16-split copies of the head are not
        rebuilt after an optimizer step - TrackingNet.refresh_head_device)"""
        prev, self.mlp = self.mlp, 'f32'
        try:
            yield
        finally:
            self.mlp = prev

    def _side_stream(self, dev):
        key = str(dev)
        st = self._side.get(key)
        if st is None:
            st = self._side[key] = torch.cuda.Stream(device=dev)
        return st

    def _part(self, tiles, N):
        return self.buf('part', tiles.T, 2, N)

    def _stash(self, key, t):
        if self.keep is not None:
            self.keep[key] = t.detach().clone()

    # ---- image branch: VGG16-BN trunk + SkipPool heads ---------------------
    def appearance(self, plan, crops, cat):
        """crops [Lt,3,S,S] NCHW (reference contract) -> cat[:, 0:512]."""
        ops, Lt, S = self.ops, plan.Lt, plan.S
        if S < 32 or S % 2 != 0:
            # five 2x2 poolings: a 32-pixel crop ends in a 1 x 1 map.  Sides that are not a multiple of 32 give odd maps
            # on the way down, floored by every pooling like nn.MaxPool2d(2, 2) (reference modules/vgg.py:72) - e.g.
            # 100 -> 50 -> 25 -> 12 -> 6 -> 3.  The first layer (NCHW crops, fused conv1_1 + conv1_2 + pool) wants an
            # even side; the reference's dataset resizes every crop to 224 (dataset/test_seq_dataset.py:218).
            raise ValueError('crop side %d is not supported: the HIP VGG trunk needs an even side >= 32; resize the crops '
                             '(mmmot_amd.crops.crop_resize_normalize)' % S)
        if Lt * S * S * 16 >= 2 ** 31 - 64:
            # the trunk kernels address activations with 32-bit offsets in 16-byte pieces (largest tensor: L x S x S x 64)
            raise ValueError('%d crops of %dx%d in one launch sequence exceed the 32-bit piece offsets of the trunk kernels '
                             '(L*S*S*16 < 2^31): split the batch' % (Lt, S, S))
        x, H, W = crops, S, S
        u8 = crops.dtype == torch.uint8  # the 8-bit crops of the resize [Lt][S][S][3]: ToTensor + Normalize on the device
        # q8: activations travel as hq8 records (fp16 hi + two e4m3 copies, same bytes)
        q8 = (self.trunk == 'f16q8') and S >= self.q8_min_crop
        f16 = self.trunk in ('f16x3', 'f16q8')  # activations travel in the hl16 split-half / hq8 format (same bytes)
        vgg = self.P['vgg']
        # conv1_1 + conv1_2 + pool as one launch when the trunk has the VGG16 head (3 -> 64 -> 64, pool)
        fuse1 = (f16 and self.fuse_conv1 and len(vgg) > 1 and vgg[0]['cout'] == 64 and
                 vgg[1]['cin'] == 64 and vgg[1]['cout'] == 64 and vgg[1]['pool'] and not vgg[0]['last'] and
                 not vgg[0]['pool'])
        if u8 and not fuse1:
            # exact-fp32 trunk / unfused first layer: they read the fp32 model input - made here from the bytes with the
            # same IEEE arithmetic as the host pipeline (one small kernel; the fused first launch takes the bytes itself)
            x = self.buf('vgg_crops32', Lt, 3, S, S)
            ops.u8_normalize(crops, self._mean_std(crops.device), x, Lt, S)
            u8 = False
        fmt = 'raw'  # format of x: 'raw' NCHW crops, 'f32' NHWC fp32, 'hl16', 'hq8' (same bytes per value)
        for li, cv in enumerate(vgg):
            if fuse1 and li == 0:
                continue
            lq8 = q8 and (self.q8_layers is None or li in self.q8_layers)  # this layer's arithmetic
            want = 'hq8' if lq8 else 'hl16'
            if f16 and fmt in ('hl16', 'hq8') and fmt != want:
                # arithmetic boundary inside an f16q8 trunk: re-encode the activation tensor (exact up to the target
                # format's own rounding)
                n = Lt * H * W * cv['cin']
                tmp = self.buf('vgg_recode32', n)
                (ops.hq8_unpack if fmt == 'hq8' else ops.hl16_unpack)(x, tmp)
                x = self.buf('vgg_recode', n)
                (ops.hq8_pack if want == 'hq8' else ops.hl16_pack)(tmp, x)
                fmt = want
            Ho, Wo = (H // 2, W // 2) if cv['pool'] else (H, W)
            out = self.buf('vgg%d' % (li & 1), Lt * Ho * Wo, cv['cout'])
            if self.conv_events is not None:  # bench.py: HIP events around every trunk launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if fuse1 and li == 1:
                c0 = vgg[0]
                if u8:
                    from .crops import MEAN, STD
                    ops.conv1_fused_u8(x, MEAN, STD, c0['wp16'], c0['bias'], c0['oscale'], cv['wpq8'] if lq8 else cv['wp16'],
                                       cv['bias'], cv['oscale'], out, Lt, H, W, q8=lq8)
                elif lq8:
                    ops.conv1_fused_hq8(x, c0['wp16'], c0['bias'], c0['oscale'], cv['wpq8'], cv['bias'], cv['oscale'],
                                        out, Lt, H, W)
                else:
                    ops.conv1_fused_hl16(x, c0['wp16'], c0['bias'], c0['oscale'], cv['wp16'], cv['bias'],
                                         cv['oscale'], out, Lt, H, W)
                fmt = want
            elif f16 and li == 0:  # unfused first layer: exact fp32 MFMA on the crops, output encoded for layer 1
                nq8 = q8 and (self.q8_layers is None or 1 in self.q8_layers)
                if nq8:
                    tmp = self.buf('vgg_first32', Lt * H * W, cv['cout'])
                    ops.conv3x3(x, cv['wp'], cv['bias'], tmp, Lt, H, W, cv['cin'], cv['cout'], True, cv['pool'])
                    ops.hq8_pack(tmp, out)
                else:
                    ops.conv3x3_first_hl16(x, cv['wp'], cv['bias'], out, Lt, H, W, cv['cout'])
                fmt = 'hq8' if nq8 else 'hl16'
            elif lq8:
                ops.conv3x3_hq8(x, cv['wpq8'], cv['bias'], out, Lt, H, W, cv['cin'], cv['cout'], cv['pool'], cv['oscale'])
                fmt = 'hq8'
            elif not f16:
                ops.conv3x3(x, cv['wp'], cv['bias'], out, Lt, H, W, cv['cin'], cv['cout'], li == 0, cv['pool'])
                fmt = 'f32'
            else:
                ops.conv3x3_hl16_patch(x, cv['wp16'], cv['bias'], out, Lt, H, W, cv['cin'], cv['cout'], cv['pool'],
                                       cv['oscale'])
                fmt = 'hl16'
            if self.conv_events is not None:
                e1.record()
                self.conv_events.append((li, Lt * H * W, cv['cin'], cv['cout'], e0, e1))
            x, H, W = out, Ho, Wo
            if cv['last']:
                self._stash('vgg_stage%d' % cv['stage'], x)
                self._skippool(plan, cv['stage'], x, H * W, cv['cout'], cat, hl16={'hq8': 2, 'hl16': 1, 'f32': 0}[fmt])

    def _mean_std(self, dev):
        key = ('mean_std', str(dev))
        if key not in self.ws:
            from .crops import MEAN, STD
            self.ws[key] = torch.tensor(list(MEAN) + list(STD), dtype=torch.float32, device=dev)
        return self.ws[key]

    def trunk_elements(self, plan):
        """activation elements the trunk writes per forward (the denominator of the range guard's fractions)"""
        n, H = 0, plan.S
        for cv in self.P['vgg']:
            if cv['pool']:
                H //= 2
            n += plan.Lt * H * H * cv['cout']
        return n

    # ---- range guard ----------------------------------------------------------
    def _range_block(self, dev):
        """this engine's own counter block (int32 [4] on `dev`) + its host mirror"""
        dev = torch.device(dev)
        if self._range_buf is None or self._range_buf.device != dev:
            self._range_buf = torch.zeros(4, dtype=torch.int32, device=dev)
            self._range_host = torch.zeros(4, dtype=torch.int32)
            if dev.type == 'cuda':
                self._range_host = self._range_host.pin_memory()
            self._range_pending = None
            self._range_seen = [0, 0, 0]  # counter values already accounted for (the block is cumulative)
        return self._range_buf

    def read_range(self, reset=True):
        """Synchronous read of this engine's counters: (e4m3-saturated, fp16-clamped, conv1_1 hits) since the last
        reset.  One stream synchronisation; GraphedForward.check_range() is the caller for captured forwards."""
        if self._range_buf is None:
            return 0, 0, 0
        v = [int(x) & 0xFFFFFFFF for x in self._range_buf.cpu().tolist()[:3]]
        d = [(a - b) & 0xFFFFFFFF for a, b in zip(v, self._range_seen)]
        if reset:
            self._range_seen = v
        return d[0], d[1], d[2]

    def _range_verdict(self, plan, sat, clamp, c11, window):
        """arithmetic the counters call for (None: the current one holds)"""
        q8 = self.trunk == 'f16q8' and plan.S >= self.q8_min_crop
        if clamp > 0 or (c11 > 0 and not q8):
            return 'f32'
        if q8 and (sat > self.q8_sat_limit * self.trunk_elements(plan) * window or c11 > 0):
            return 'f16x3'
        return None

    def _range_event(self, plan, lower, sat, clamp, c11, recomputed, window=None):
        # `affected_forwards` = (first, last): indices (count of forwards of this engine, from 0) of the forwards whose
        # trunk launches the counters that tripped cover.  recomputed=True: one forward, the current one, and its results
        # were recomputed in the lowered arithmetic before they were returned.  False (asynchronous read-back): a read-back
        # is recorded behind forward k and inspected once its copy has completed - possibly several forwards later, and no
        # new one is queued meanwhile - so it covers every forward since the previous read-back up to k, NOT "the previous
        # forward": `window` carries exactly that range, stored with the pending copy when it was recorded.
        # Those results were ALREADY RETURNED: the caller discards / repeats the forwards first .. last
        # (`Engine.forward_index` of a result = the value of `_n_forward - 1` after the call; `last_out_of_range_forward`
        # = last and `out_of_range_window` = (first, last) keep the latest detection, None while there was none).
        # `affected_forward` (the last index of the range) is kept for callers of the round-3 interface.
        if recomputed or window is None:
            window = (self._n_forward, self._n_forward)
        ev = dict(forward=self._n_forward, affected_forward=window[1], affected_forwards=tuple(window), was=self.trunk,
                  now=lower, e4m3_saturated=sat, fp16_clamped=clamp, conv1_1_hits=c11,
                  trunk_elements=self.trunk_elements(plan), recomputed=recomputed)
        self.range_events.append(ev)
        if not recomputed:
            self.last_out_of_range_forward = window[1]
            self.out_of_range_window = tuple(window)
        warnings.warn('mmmot_amd range guard: trunk arithmetic %(was)s -> %(now)s (%(e4m3_saturated)d activation '
                      'elements beyond the e4m3 range, %(fp16_clamped)d beyond the fp16 range, of %(trunk_elements)d '
                      'per forward); ' % ev + ('the trunk of this forward is recomputed' if recomputed else
                                               'detected late: forwards %d..%d of this engine ran out of range and were '
                                               'already returned' % tuple(window)),
                      RuntimeWarning, stacklevel=4)
        self.trunk = lower

    def _guarded_appearance(self, plan, crops, cat):
        """appearance() under the range guard (see __init__).  Every engine owns its counter block (bound around its
        trunk launches: mmmot_trunk_range_bind), so engines sharing a device and captured graphs never mix windows.
        First forward: synchronous check, the trunk is recomputed in the lowered arithmetic when it trips.  Every
        later forward: the 16-byte block is copied to pinned host memory behind the trunk (asynchronous) and inspected
        at the start of the NEXT forward - no synchronisation, detection lags by one step (the event says so);
        `range_check_every` > 0 adds a synchronous check-and-recompute every that many forwards.  Never inside a
        hipGraph capture: GraphedForward.check_range() reads the block of a captured forward."""
        ops = self.ops
        dev = crops.device
        capturing = torch.cuda.is_current_stream_capturing() if crops.is_cuda else False
        active = self.range_guard and hasattr(ops, 'trunk_range_bind')
        if not active:
            self.appearance(plan, crops, cat)
            self._n_forward += 1
            return
        blk = self._range_block(dev)
        first = self._n_forward == 0
        guard = self.trunk != 'f32' and not capturing
        # ---- verdict of the previous forward's asynchronous read-back -------------------------------------------
        if guard and self._range_pending is not None:
            ev, w0, w1 = self._range_pending
            if ev is True or ev.query():
                self._range_pending = None
                v = [int(x) & 0xFFFFFFFF for x in self._range_host.tolist()[:3]]
                sat, clamp, c11 = [(a - b) & 0xFFFFFFFF for a, b in zip(v, self._range_seen)]
                self._range_seen = v
                lower = self._range_verdict(plan, sat, clamp, c11, w1 - w0 + 1)
                if lower is not None:
                    self._range_event(plan, lower, sat, clamp, c11, recomputed=False, window=(w0, w1))
                    guard = self.trunk != 'f32'
        sync = guard and (first or (self.range_check_every > 0 and self._n_forward % self.range_check_every == 0))
        if sync:
            # open the window of this forward.  What the counters hold at this point belongs to the forwards since the last
            # inspected read-back (an asynchronous copy that had not completed when this check came round is superseded
            # by this synchronous read): they get their verdict and event here instead of being dropped
            sat, clamp, c11 = self.read_range(reset=True)
            w0, w1 = self._range_win0, self._n_forward - 1
            if self._range_pending is not None:  # the copy that did not complete: its forwards are in these counters too
                w0 = min(w0, self._range_pending[1])
            self._range_pending = None
            if w1 >= w0:
                lower = self._range_verdict(plan, sat, clamp, c11, w1 - w0 + 1)
                if lower is not None:
                    self._range_event(plan, lower, sat, clamp, c11, recomputed=False, window=(w0, w1))
                    sync = self.trunk != 'f32'  # lowered to the exact arithmetic: nothing left to check
            self._range_win0 = self._n_forward
        ops.trunk_range_bind(blk)
        try:
            self.appearance(plan, crops, cat)
            while sync and self.trunk != 'f32':
                sat, clamp, c11 = self.read_range(reset=True)
                lower = self._range_verdict(plan, sat, clamp, c11, 1)
                if lower is None:
                    break
                self._range_event(plan, lower, sat, clamp, c11, recomputed=True)
                self.appearance(plan, crops, cat)
        finally:
            ops.trunk_range_bind(None)
        if guard and not sync and self.trunk != 'f32' and self._range_pending is None:
            # stream-ordered: copy the (cumulative) block out and mark the point - inspected by the next forward
            self._range_host.copy_(blk, non_blocking=True)
            e = True
            if dev.type == 'cuda':
                e = torch.cuda.Event()
                e.record()
            # the counters are cumulative: this copy covers the forwards since the previous read-back up to this one
            self._range_pending = (e, self._range_win0, self._n_forward)
            self._range_win0 = self._n_forward + 1
        elif sync:
            self._range_win0 = self._n_forward + 1  # checked (and, if need be, recomputed) synchronously
        self._n_forward += 1

    def _skippool(self, plan, s, x, hw, C, cat, hl16=False):
        """reference modules/appear_net.py:9-32 for stage s -> cat[:, 128 s : 128 (s+1)]."""
        ops, Lt, hd, T = self.ops, plan.Lt, self.P['skippool'][s], plan.det_tiles
        pooled = self.buf('sp_pool', Lt, C)
        first, second, npart = plan.crop_segments(hw)
        if second is None:
            ops.segment_mean(x, C, first, pooled, use_group=False, hl16=hl16)
        else:  # two-level pool: row chunks of a crop -> partial sums -> mean (few crops: see plan._crop_segments)
            partial = self.buf('sp_partial', npart, C)
            ops.segment_mean(x, C, first, partial, use_group=False, hl16=hl16)
            ops.segment_mean(partial, C, second, pooled, use_group=False)
        if self.sp_fused and hasattr(ops, 'skippool_head'):
            ops.skippool_head(pooled, C, hd, EPS, cat[:, 128 * s:128 * (s + 1)], Lt)  # the whole head in one launch
            return
        ln0 = self.buf('sp_ln0', Lt, C)
        ops.row_layernorm(pooled, C, hd['g0'], hd['b0'], EPS, False, ln0, Lt)
        C4 = hd['w1'].shape[0]
        h1 = self.buf('sp_h1', Lt, C4)
        self._gemm(hd, 'w1', T, C4, C, X=ln0, bias=hd['c1'], Y=h1)
        ln1 = self.buf('sp_ln1', Lt, C4)
        ops.row_layernorm(h1, C4, hd['g2'], hd['b2'], EPS, True, ln1, Lt)
        h2 = self.buf('sp_h2', Lt, 128)
        self._gemm(hd, 'w4', T, 128, C4, X=ln1, bias=hd['c4'], Y=h2)
        ops.row_layernorm(h2, 128, hd['g5'], hd['b5'], EPS, True, cat[:, 128 * s:128 * (s + 1)], Lt)

    # ---- LiDAR branch: PointNet with folded transforms ----------------------
    def pointnet(self, plan, points, cat):
        """points [P,3] -> cat[:, 512:1024]; reference modules/point_net.py:25-44,115-153."""
        ops, pn, T, D, Pn, Lt = self.ops, self.P['pointnet'], plan.pt_tiles, plan.det_tiles, plan.P, plan.Lt
        y1 = self.buf('pn_y1', Pn, 64)
        part = self._part(T, 64)
        ops.pointnet_layer1(points, pn['w1'], pn['b1'], y1, part, T)
        sc1, sh1 = self._finalize('pn1', part, T, 64, 64, pn['g1'], pn['be1'])
        # conv2..conv5: each consumes relu(gn(previous)) through the GEMM prologue
        fused = self.pn_fused
        TD = plan.ptd_tiles if fused else T  # detection-aligned tiles for the fused epilogues
        x, sc, sh = y1, sc1, sh1
        for i, (N, K) in zip((2, 3, 4), ((64, 64), (64, 64), (128, 64))):
            y = self.buf('pn_y%d' % i, Pn, N)
            part = self._part(T, N)
            if self.pn_mlp64 and self.mlp == 'f16x3' and ('w%d_h16' % i) in pn:
                ops.pn_mlp64(pn['w%d_h16' % i], pn['w%d_os' % i], T, N, x, sc, sh, pn['b%d' % i], y, part)
            else:
                self._gemm(pn, 'w%d' % i, T, N, K, X=x, bias=pn['b%d' % i], Y=y, part=part, sc=sc, sh=sh,
                           amode=A_NORM_RELU)
            sc, sh = self._finalize('pn%d' % i, part, T, N, N, pn['g%d' % i], pn['be%d' % i])
            x = y
        # conv5 128->1024 + GN + ReLU + per-detection average (named max_feats in the reference,
        # point_net.py:138-148).  Fused: the [P][1024] tensor (1 GiB per cfg3 pair) is never stored - one GEMM
        # pass for the statistics, one that normalises in the epilogue and emits per-tile column sums.
        seg1024 = self.buf('pn_seg1024', Lt, 1024)
        ares = fused and self.mlp == 'f16x3' and 'w5_h16' in pn  # A-resident kernel (hl16 weights only)
        TH = plan.ptd_half if ares else None
        if ares:
            if self.pn_gram:
                # statistics of conv5's output from the second moments of its 128-channel input: no GEMM pass
                GT = plan.gram_tiles
                Gp, Sp = self.buf64('gram_G', GT.T, 128 * 128), self.buf64('gram_S', GT.T, 128)
                ops.gram_rows(x, 128, sc, sh, GT, Gp, Sp)
                sc5, sh5 = self.buf('pn5_sc', GT.G, 1024), self.buf('pn5_sh', GT.G, 1024)
                ops.gn_finalize_gram(Gp, Sp, GT, 128, pn['w5'], pn['b5'], 1024, pn['g5'], pn['be5'], EPS,
                                     self.buf64('gram_work', GT.G, 128 * 128 + 128), sc5, sh5)
            else:
                part = self._part(TH, 1024)
                ops.gemm_ares(pn['w5_h16'], pn['w5_os'], TD, 1024, 128, x, sc, sh, bias=pn['b5'], part=part)
                sc5, sh5 = self._finalize('pn5', part, TH, 1024, 1024, pn['g5'], pn['be5'])
            cs = self.buf('pn_colsum', TH.T, 1024)
            if self.conv_events is not None:  # bench.py: HIP events around PointNet's dominant launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.gemm_ares(pn['w5_h16'], pn['w5_os'], TD, 1024, 128, x, sc, sh, bias=pn['b5'], osc=sc5, osh=sh5,
                          colsum=cs)
            if self.conv_events is not None:
                e1.record()
                self.conv_events.append(('pn5', Pn, 128, 1024, e0, e1))
            ops.segment_mean(cs, 1024, plan.det_half_segs, seg1024)
        elif fused:
            part = self._part(TD, 1024)
            self._gemm(pn, 'w5', TD, 1024, 128, X=x, bias=pn['b5'], part=part, sc=sc, sh=sh, amode=A_NORM_RELU)
            sc5, sh5 = self._finalize('pn5', part, TD, 1024, 1024, pn['g5'], pn['be5'])
            cs = self.buf('pn_colsum', TD.T, 1024)
            self._gemm(pn, 'w5', TD, 1024, 128, X=x, bias=pn['b5'], sc=sc, sh=sh, amode=A_NORM_RELU,
                       osc=sc5, osh=sh5, colsum=cs)
            ops.segment_mean(cs, 1024, plan.det_tile_segs, seg1024)
        else:
            y = self.buf('pn_y5', Pn, 1024)
            part = self._part(T, 1024)
            self._gemm(pn, 'w5', T, 1024, 128, X=x, bias=pn['b5'], Y=y, part=part, sc=sc, sh=sh, amode=A_NORM_RELU)
            sc5, sh5 = self._finalize('pn5', part, T, 1024, 1024, pn['g5'], pn['be5'])
            ops.segment_mean(y, 1024, plan.det_segs, seg1024, sc=sc5, sh=sh5, relu=True)
        self._stash('pn_seg1024', seg1024)
        # PointNet_v1.conv1 split: per-detection 1024-channel part becomes a gathered bias
        dbias = self.buf('pn_dbias', Lt, 512)
        self._gemm(pn, 'wc1b', D, 512, 1024, X=seg1024, bias=pn['bc1'], Y=dbias)
        seg512 = self.buf('pn_seg512', Lt, 512)
        if ares:
            if self.pn_gram and hasattr(ops, 'gn_finalize_gram_dbias'):
                # statistics of v = W a + dbias[det] from the second moments of the 64-channel input a = relu(gn(y1)) over
                # detection-aligned super-tiles: a pass over 256 B per point instead of the 64 -> 512 GEMM's statistics pass
                GT = plan.gram64_tiles
                Gp, Sp = self.buf64('gram64_G', GT.T, 64 * 64), self.buf64('gram64_S', GT.T, 64)
                ops.gram_rows(y1, 64, sc1, sh1, GT, Gp, Sp)
                scc, shc = self.buf('pnc1_sc', GT.G, 512), self.buf('pnc1_sh', GT.G, 512)
                ops.gn_finalize_gram_dbias(Gp, Sp, GT, plan.gram64_tile_det, 64, pn['wc1a'], dbias, 512, pn['gc1'], pn['bec1'],
                                           EPS, self.buf64('gram64_work', GT.G, 64 * 64 + 64), scc, shc)
            else:
                part = self._part(TH, 512)
                ops.gemm_ares(pn['wc1a_h16'], pn['wc1a_os'], TD, 512, 64, y1, sc1, sh1, dbias=dbias,
                              tile_dbrow=plan.tile_det, part=part)
                scc, shc = self._finalize('pnc1', part, TH, 512, 512, pn['gc1'], pn['bec1'])
            cs = self.buf('pn_colsum', TH.T, 512)
            ops.gemm_ares(pn['wc1a_h16'], pn['wc1a_os'], TD, 512, 64, y1, sc1, sh1, dbias=dbias,
                          tile_dbrow=plan.tile_det, osc=scc, osh=shc, colsum=cs)
            ops.segment_mean(cs, 512, plan.det_half_segs, seg512)
        elif fused:
            part = self._part(TD, 512)
            self._gemm(pn, 'wc1a', TD, 512, 64, X=y1, part=part, sc=sc1, sh=sh1, amode=A_NORM_RELU,
                       dbias=dbias, rowidx=plan.row_det)
            scc, shc = self._finalize('pnc1', part, TD, 512, 512, pn['gc1'], pn['bec1'])
            cs = self.buf('pn_colsum', TD.T, 512)
            self._gemm(pn, 'wc1a', TD, 512, 64, X=y1, sc=sc1, sh=sh1, amode=A_NORM_RELU, dbias=dbias,
                       rowidx=plan.row_det, osc=scc, osh=shc, colsum=cs)
            ops.segment_mean(cs, 512, plan.det_tile_segs, seg512)
        else:
            yc1 = self.buf('pn_yc1', Pn, 512)
            part = self._part(T, 512)
            self._gemm(pn, 'wc1a', T, 512, 64, X=y1, Y=yc1, part=part, sc=sc1, sh=sh1, amode=A_NORM_RELU,
                       dbias=dbias, rowidx=plan.row_det)
            scc, shc = self._finalize('pnc1', part, T, 512, 512, pn['gc1'], pn['bec1'])
            ops.segment_mean(yc1, 512, plan.det_segs, seg512, sc=scc, sh=shc, relu=True)
        yc2 = self.buf('pn_yc2', Lt, 512)
        part = self._part(D, 512)
        self._gemm(pn, 'wc2', D, 512, 512, X=seg512, bias=pn['bc2'], Y=yc2, part=part)
        sc2, sh2 = self._finalize('pnc2', part, D, 512, 16, pn['gc2'], pn['bec2'])
        ops.affine_act(yc2, 512, sc2, sh2, D, ACT_RELU, cat[:, 512:1024])

    # ---- fusion module A / B / C --------------------------------------------
    def fuse(self, plan, cat, F):
        """cat [Lt,1024] -> F [3,Lt,512]; reference modules/fusion_net.py."""
        ops, fu, D, Lt = self.ops, self.P['fusion'], plan.det_tiles, plan.Lt
        mode = FUSION_MODES[self.fusion]
        img, pts = cat[:, 0:512], cat[:, 512:1024]
        if self.fusion == 'A':
            y0 = self.buf('fu_y0', Lt, 512)
            part = self._part(D, 512)
            self._gemm(fu, 'w0', D, 512, 1024, X=cat, bias=fu['b0'], Y=y0, part=part)
            sc0, sh0 = self._finalize('fu0', part, D, 512, 512, fu['g0'], fu['be0'])
            ops.fusion_combine(mode, cat, y0, None, sc0, sh0, None, None, D, F, Lt, 512)
            return
        N = 512 if self.fusion == 'B' else 1024
        ys, scs, shs = [], [], []
        for j, x in enumerate((img, pts)):  # NB: *_p weights consume the IMAGE features
            y = self.buf('fu_y%d' % j, Lt, N)
            part = self._part(D, N)
            self._gemm(fu, 'w%d' % j, D, N, 512, X=x, bias=fu['b%d' % j], Y=y, part=part)
            sc, sh = self._finalize('fu%d' % j, part, D, N, N, fu['g%d' % j], fu['be%d' % j])
            ys.append(y)
            scs.append(sc[:, N - 512:])
            shs.append(sh[:, N - 512:])
        ops.fusion_combine(mode, cat, ys[0], ys[1], scs[0], shs[0], scs[1], shs[1], D, F, Lt, 512)

    # ---- negative-rejection head --------------------------------------------
    def det_scores(self, plan, F):
        """F [nR,Lt,512] -> [nR,Lt]; reference modules/tracking_net.py:91-100,149-163 (eval)."""
        ops, wd, T = self.ops, self.P['w_det'], plan.F_tiles
        R = plan.nR * plan.Lt
        X = F.view(R, 512)
        h0 = self.buf('det_h0', R, 512)
        self._gemm(wd, 'w0', T, 512, 512, X=X, bias=wd['b0'], Y=h0, act=ACT_RELU)
        h1 = self.buf('det_h1', R, 256)
        self._gemm(wd, 'w3', T, 256, 512, X=h0, bias=wd['b3'], Y=h1, act=ACT_RELU)
        out = torch.empty(plan.nR, plan.Lt, dtype=torch.float32, device=F.device)
        act = ACT_SIGMOID if 'cls' in self.score_arch else ACT_NONE
        ops.rowdot(h1, 256, wd['w6'], wd['b6'], T, out.view(-1), act=act, use_thr=True, thr=self.neg_threshold)
        return out

    # ---- pairwise affinity + new/end + softmax ------------------------------
    def affinity(self, plan, F):
        """reference modules/gcn.py:68-82, new_end.py:62-82, tracking_net.py:106-126."""
        ops, lk, PT, VT = self.ops, self.P['w_link'], plan.pair_tiles, plan.v_tiles
        nR, Lt, R = plan.nR, plan.Lt, plan.pair_tiles.R
        Ff = F.view(nR * Lt, 512)
        # uniform32 (include/mmmot_hip.h: pair_uniform32): every 32-row block of a pair tile lies inside one i - needs
        # M % 32 == 0 in every group AND tiles that start a multiple of 32 rows into their group
        pair = dict(row0=PT.g_row0, M=plan.pg_M, aoff=plan.pg_aoff, boff=plan.pg_boff, uniform32=plan.pair_uniform32)
        # stacked [new_end.conv0 ; conv1.0] over the on-the-fly pairwise tensor
        ya = self.buf('aff_ya', R, 1024)
        part = self._part(PT, 1024)
        self._gemm(lk, 'wa', PT, 1024, 512, FA=Ff, FB=Ff, pair=pair, amode=A_PAIR,
                 pairop=PAIR_OPS[self.affinity_op], bias=lk['ba'], Y=ya, part=part)
        sc_ne, sh_ne = self._finalize('aff_ne0', part[:, :, 0:512], PT, 512, 1, lk['g_ne0'], lk['be_ne0'])
        sc1, sh1 = self._finalize('aff_1', part[:, :, 512:1024], PT, 512, 512, lk['g1'], lk['be1'])
        # new / end vectors: strided means of relu(gn(conv0)) over the prev / curr axis
        V = self.buf('aff_v', VT.R, 512)
        ops.segment_mean(ya[:, 0:512], 512, plan.v_segs, V, sc=sc_ne, sh=sh_ne, relu=True,
                         take_max=(self.end_mode != 'avg'))
        self._stash('aff_v', V)
        vh0 = self.buf('aff_vh0', VT.R, 512)
        part = self._part(VT, 512)
        self._gemm(lk, 'nw0', VT, 512, 512, X=V, bias=lk['nb0'], Y=vh0, part=part)
        scv, shv = self._finalize('aff_v1', part, VT, 512, 1, lk['ng1'], lk['nbe1'])
        vh1 = self.buf('aff_vh1', VT.R, 128)
        part = self._part(VT, 128)
        self._gemm(lk, 'nw3', VT, 128, 512, X=vh0, bias=lk['nb3'], Y=vh1, part=part, sc=scv, sh=shv,
                 amode=A_NORM_RELU)
        scv2, shv2 = self._finalize('aff_v4', part, VT, 128, 1, lk['ng4'], lk['nbe4'])
        ne = torch.zeros(2, nR, Lt, dtype=torch.float32, device=F.device)  # eval-mode zero padding (tracking_net.py:183-189)
        ops.rowdot(vh1, 128, lk['nw6'], lk['nb6'], VT, ne.view(-1), sc=scv2, sh=shv2, act=ACT_SIGMOID,
                   omap=plan.v_omap)
        # link branch
        y3 = self.buf('aff_y3', R, 512)
        part = self._part(PT, 512)
        self._gemm(lk, 'w3', PT, 512, 512, X=ya[:, 512:1024], bias=lk['b3'], Y=y3, part=part, sc=sc1, sh=sh1,
                 amode=A_NORM_RELU)
        sc4, sh4 = self._finalize('aff_4', part, PT, 512, 512, lk['g4'], lk['be4'])
        y6 = self.buf('aff_y6', R, 128)
        part = self._part(PT, 128)
        self._gemm(lk, 'w6', PT, 128, 512, X=y3, bias=lk['b6'], Y=y6, part=part, sc=sc4, sh=sh4,
                 amode=A_NORM_RELU)
        sc7, sh7 = self._finalize('aff_7', part, PT, 128, 128, lk['g7'], lk['be7'])
        logits = torch.empty(R, dtype=torch.float32, device=F.device)
        ops.rowdot(y6, 128, lk['w9'], lk['b9'], PT, logits, sc=sc7, sh=sh7)
        link = logits
        if self.softmax_mode != 'none':
            link = torch.empty_like(logits)
            ops.softmax_pairs(logits, link, PT.g_row0, plan.pg_N, plan.pg_M, PT.G, plan.max_nm,
                              SOFTMAX_MODES[self.softmax_mode])
        return link, ne[0], ne[1]

    # ---- whole forward -------------------------------------------------------
    def forward(self, plan, crops=None, points=None):
        """Returns dict(det [nR,Lt], link flat, new [nR,Lt], end [nR,Lt], feats F, cat)."""
        # One forward at a time per engine: the workspace arena (self.ws), the packed weights and the range-guard
        # block are mutable state behind the integer handle of the registered operators (mmmot_amd/torch_ops.py), which
        # an operator schema cannot express.  A second thread entering while a forward is being issued is refused; a
        # forward issued on ANOTHER stream than the previous one first waits for that stream (the previous forward's
        # kernels still read and write the same workspace), so consecutive forwards are ordered whatever stream they
        # are launched on.  Outputs: det / link / new / end are freshly allocated per call; 'F' and 'cat' are views
        # of the workspace, valid until the next forward of this engine.
        if not self._busy.acquire(blocking=False):
            raise RuntimeError('mmmot_amd: concurrent forwards on one engine (its workspace arena is shared mutable state); '
                               'use one TrackingNet per thread, or serialise the calls')
        try:
            pin = getattr(self.ops, 'on_current_stream', None)
            if pin is None:
                return self._forward(plan, crops, points)
            ref = crops if crops is not None else points
            cur = None
            if ref is not None and ref.is_cuda:
                cur = torch.cuda.current_stream(ref.device)
                last = self._last_stream
                if last is not None and last != cur and not torch.cuda.is_current_stream_capturing():
                    cur.wait_stream(last)
                self._last_stream = cur
            with pin(cur):
                return self._forward(plan, crops, points)
        finally:
            self._busy.release()

    def _check_crops(self, plan, crops):
        Lt = plan.Lt
        ok_f32 = crops is not None and crops.dtype == torch.float32 and tuple(crops.shape) == (Lt, 3, plan.S, plan.S)
        ok_u8 = crops is not None and crops.dtype == torch.uint8 and tuple(crops.shape) == (Lt, plan.S, plan.S, 3)
        if not (ok_f32 or ok_u8) or not crops.is_contiguous():
            raise ValueError('crops must be a contiguous fp32 [%d,3,%d,%d] tensor (the reference\'s normalised `dets`) or '
                             'the uint8 [%d,%d,%d,3] crops of the resize' % (Lt, plan.S, plan.S, Lt, plan.S, plan.S))

    def image_first(self, plan, crops):
        """Issue the image branch (trunk + SkipPool heads -> the appearance half of `cat`) of the NEXT ``forward`` now.
        The trunk needs the detections' counts only, not the point split: the reference-shaped call (TrackingNet.forward)
        launches it before it reads ``points_split`` back and builds the full plan, so that ~0.3 ms of host work hide
        behind ~2 ms of device work.  `plan` may be any plan with the same frame counts and crop side (its image tables are
        the ones used); the next ``forward`` with the same crops tensor skips its own image branch."""
        if not self._busy.acquire(blocking=False):
            raise RuntimeError('mmmot_amd: concurrent forwards on one engine (its workspace arena is shared mutable state); '
                               'use one TrackingNet per thread, or serialise the calls')
        try:
            self._check_crops(plan, crops)
            pin = getattr(self.ops, 'on_current_stream', None)
            cur = None
            if crops.is_cuda:
                cur = torch.cuda.current_stream(crops.device)
                last = self._last_stream
                if last is not None and last != cur and not torch.cuda.is_current_stream_capturing():
                    cur.wait_stream(last)
                self._last_stream = cur
            self.dev = crops.device
            cat = self.buf('cat', plan.Lt, 1024)
            # what was queued before the trunk: the point the LiDAR branch of the coming forward may start from (side stream)
            self._pre_image = None
            if cur is not None and self.pn_beside_trunk and not torch.cuda.is_current_stream_capturing():
                self._pre_image = torch.cuda.Event()
                self._pre_image.record(cur)
            if pin is None:
                self._guarded_appearance(plan, crops, cat)
            else:
                with pin(cur):
                    self._guarded_appearance(plan, crops, cat)
            self._image_token = (crops.data_ptr(), plan.Lt, plan.S, crops.dtype)
        finally:
            self._busy.release()

    def _forward(self, plan, crops=None, points=None):
        rows = plan.rows
        need_img = (0 in rows) or (2 in rows)
        need_pts = (1 in rows) or (2 in rows)
        dev = crops.device if crops is not None else points.device
        self.dev = dev
        Lt = plan.Lt
        cat = self.buf('cat', Lt, 1024)
        token, self._image_token = self._image_token, None
        if need_img:
            self._check_crops(plan, crops)
        # image_first() ran the image branch of exactly these crops into `cat` already
        img_done = need_img and token is not None and token == (crops.data_ptr(), Lt, plan.S, crops.dtype)
        if need_pts:
            kin = int(self.P['pointnet']['w1'].shape[1])  # 3 (xyz) or 4 (xyz + reflectivity)
            if points is None or tuple(points.shape) != (plan.P, kin) or not points.is_contiguous():
                raise ValueError('points must be a contiguous [%d,%d] tensor' % (plan.P, kin))
        # The two branches meet only in `cat` (disjoint column halves, disjoint workspace buffers), so the LiDAR branch
        # can run on a side stream (two_streams, opt-in): the trunk's persistent workgroups fill every CU (one per CU,
        # all its LDS and registers) - the branches never share a CU, only PointNet's small-grid launches and the
        # uneven tail of a trunk layer leave CUs to the other stream.  Fork / join are events: capturable in a hipGraph.
        side = None
        pre, self._pre_image = self._pre_image, None
        if (img_done and need_pts and pre is not None and dev.type == 'cuda' and hasattr(self.ops, 'on_stream')
                and not torch.cuda.is_current_stream_capturing()):
            # the trunk of this forward is already running (image_first): the LiDAR branch goes beside it - it starts from
            # the point the trunk was launched at, not behind it.  At the reference's call shape (one frame pair) the last
            # trunk layers leave CUs idle (conv5 of 22 crops: 176 tiles on 256 CUs) and PointNet's ~15 small launches
            # hide there
            main = torch.cuda.current_stream(dev)
            sd = self._side_stream(dev)
            sd.wait_event(pre)
            with self.ops.on_stream(sd):
                self.pointnet(plan, points, cat)
            main.wait_stream(sd)
            need_pts = False  # done
        elif need_img and need_pts and self.two_streams and dev.type == 'cuda' and hasattr(self.ops, 'on_stream'):
            side = self._side_stream(dev)
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)  # inputs (and the previous forward's readers of the workspace) are ordered before
            with self.ops.on_stream(side):
                self.pointnet(plan, points, cat)
            if not img_done:
                self._guarded_appearance(plan, crops, cat)
            main.wait_stream(side)
        else:
            if need_img and not img_done:
                self._guarded_appearance(plan, crops, cat)
            if need_pts:
                self.pointnet(plan, points, cat)
        F = self.buf('F', plan.nR, Lt, 512)
        if rows == (0, 1, 2):
            self.fuse(plan, cat, F)
        else:
            for ri, r in enumerate(rows):  # single-modality rows: a device copy, no arithmetic
                if r == 2:
                    raise ValueError('the fused row needs rows=(0,1,2)')
                F[ri].copy_(cat[:, 512 * r:512 * (r + 1)])
        det = self.det_scores(plan, F)
        link, new, end = self.affinity(plan, F)
        return dict(det=det, link=link, new=new, end=end, F=F, cat=cat)
