"""In-process per-launch-class timing of the forward: HIP events around every operator call of ``HipOps``.

``bench.py`` uses it for ``extra.kernels`` (the driver's record must carry, per launch class,
ms per step, the roofline that bounds it and the achieved rate - not only builder-run rocprof tables).  It is a
measurement aid, not part of the product path: outside a ``LaunchProfiler`` block ``HipOps`` is untouched.

How it measures.  Inside the block every operator method listed in ``COSTS`` is replaced (on the ops INSTANCE) by a
wrapper that records one ``torch.cuda.Event`` before and one after the call on the stream the call launches on (the
engine pins torch's current stream for a launch sequence, and its side-stream blocks switch torch's current stream too,
so ``torch.cuda.current_stream()`` is that stream).  Kernels of one stream run back to back, so the event pair around a
call measures that call's kernels plus the (sub-microsecond) gap in front of them; the sum over a step's calls is the
step's device time.  Each class also carries its algorithmic FLOPs and its algorithmic HBM bytes (inputs read once,
outputs written once, weights once), from which the summary derives the bound (whichever of FLOPs / MFMA peak and
bytes / HBM peak is the longer time), the achieved TFLOP/s(-equivalent) or TB/s and the fraction of that roofline.
Launch classes whose roofline time is under 5 us per launch are labelled "latency": neither rate means anything there.
"""
import time

import torch

PEAK_F16X3_TFLOPS = 2500.0 / 3.0  # f16 MFMA dense peak / 3 MFMAs per algorithmic product
PEAK_F32_TFLOPS = 157.3           # fp32-input MFMA (the exact-fp32 mode)
PEAK_HBM_TBS = 8.0                # /opt/skills/guides/MI355X_MICROARCH.md


def _rows(t):
    return int(getattr(t, 'R', 0))


def _seg_rows(segs):
    return int(segs.h_count.sum())


def _conv(name):
    def cost(a, k):
        inp, wp, bias, out, L, H, W, Cin, Cout = a[:9]
        pool = a[10] if name == 'conv3x3' else a[9]
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        fl = 2.0 * L * H * W * 9 * Cin * Cout
        by = 4.0 * (L * H * W * Cin + L * Ho * Wo * Cout + 9 * Cin * Cout)
        return ('trunk conv %d->%d @%dx%d%s' % (Cin, Cout, H, W, ' +pool' if pool else ''), fl, by)
    return cost


def _conv1_fused(a, k):
    # (..., out, L, H, W[, q8]): the three integers behind the output tensor
    i = max(j for j, v in enumerate(a) if torch.is_tensor(v))
    L, H, W = a[i + 1:i + 4]
    fl = 2.0 * L * H * W * 9 * (3 * 64 + 64 * 64)
    by = (1.0 if a[0].dtype == torch.uint8 else 4.0) * L * H * W * 3 + 4.0 * L * (H // 2) * (W // 2) * 64
    return ('trunk conv1_1+conv1_2 (fused) 3->64->64 @%dx%d +pool' % (H, W), fl, by)


def _gemm(a, k):
    W, tiles, N, K = a[:4]
    R = _rows(tiles)
    amode = k.get('amode', 0)
    kind = {0: 'plain', 1: 'norm+relu prologue', 2: 'pair prologue'}[amode]
    by = 4.0 * N * K + (4.0 * R * N if k.get('Y') is not None else 0.0)
    if amode != 2 and k.get('X') is not None:
        by += 4.0 * R * K
    if k.get('colsum') is not None or (k.get('Y') is None and k.get('part') is not None):
        kind += ', reduced'
    big = 'rows GEMM' if R >= 4096 else 'small GEMM'
    return ('%s %d->%d (%s)' % (big, K, N, kind), 2.0 * R * N * K, by)


def _gemm_ares(a, k):
    W16, oscale, tiles, N, K, X = a[:6]
    R = _rows(tiles)
    what = 'column sums' if k.get('colsum') is not None else 'statistics'
    return ('A-resident GEMM %d->%d (norm+relu prologue, %s only)' % (K, N, what), 2.0 * R * N * K, 4.0 * R * K + 4.0 * N * K)


def _pn_mlp64(a, k):
    W16, oscale, tiles, N = a[:4]
    R = _rows(tiles)
    return ('PointNet layer 64->%d (pn_mlp64)' % N, 2.0 * R * 64 * N, 4.0 * R * (64 + N))


def _gram(a, k):
    X, K, sc, sh, tiles = a[:5]
    R = _rows(tiles)
    return ('Gram matrix of the %d-channel rows (conv5 statistics)' % K, 2.0 * R * K * K, 4.0 * R * K)


def _segment_mean(a, k):
    X, C, segs, out = a[:4]
    R = _seg_rows(segs)
    kind = 'hl16 rows' if k.get('hl16') else ('norm+relu rows' if k.get('sc') is not None else 'rows')
    big = 'segment mean' if R >= 65536 else 'small segment mean'
    return ('%s, %d channels (%s)' % (big, C, kind), 1.0 * R * C, 4.0 * R * C)


def _rowdot(a, k):
    X, K, w, b, tiles, out = a[:6]
    R = _rows(tiles)
    return ('row dot %d->1' % K, 2.0 * R * K, 4.0 * R * K)


def _latency(label):
    return lambda a, k: (label, 0.0, 0.0)


def _layer1(a, k):
    X, W, bias, Y, part, tiles = a[:6]
    R = _rows(tiles)
    Kin = int(W.shape[1])
    return ('PointNet layer %d->64 (points in)' % Kin, 2.0 * R * Kin * 64, 4.0 * R * (Kin + 64))


COSTS = {
    'conv3x3': _conv('conv3x3'),
    'conv3x3_hl16_patch': _conv('hl16'),
    'conv3x3_hq8': _conv('hq8'),
    'conv1_fused_hl16': _conv1_fused,
    'conv1_fused_hq8': _conv1_fused,
    'conv1_fused_u8': _conv1_fused,
    'conv3x3_first_hl16': lambda a, k: ('trunk conv1_1 3->%d (fp32 MFMA)' % a[7], 2.0 * a[4] * a[5] * a[6] * 27 * a[7],
                                        4.0 * a[4] * a[5] * a[6] * (3 + a[7])),
    'gemm': _gemm,
    'gemm_ares': _gemm_ares,
    'pn_mlp64': _pn_mlp64,
    'gram_rows': _gram,
    'gn_finalize_gram': _latency('GroupNorm scale/shift from the Gram matrix'),
    'gn_finalize_gram_dbias': _latency('GroupNorm scale/shift from the Gram matrix'),
    'gn_finalize': _latency('GroupNorm scale/shift from tile partials'),
    'segment_mean': _segment_mean,
    'rowdot': _rowdot,
    'row_layernorm': _latency('row LayerNorm'),
    'skippool_head': _latency('SkipPool head (LN, 1x1, LN, 1x1, LN)'),
    'pointnet_layer1': _layer1,
    'affine_act': _latency('affine + activation rows'),
    'fusion_combine': _latency('fusion combine'),
    'softmax_pairs': _latency('pair softmax'),
    'u8_normalize': _latency('8-bit crops -> fp32'),
    'hq8_pack': _latency('hq8 / hl16 re-encode'), 'hq8_unpack': _latency('hq8 / hl16 re-encode'),
    'hl16_pack': _latency('hq8 / hl16 re-encode'), 'hl16_unpack': _latency('hq8 / hl16 re-encode'),
}


class LaunchProfiler:
    """``with LaunchProfiler(ops) as prof: step(); step()`` then ``prof.summary(steps=2)``."""

    def __init__(self, ops, f32=False):
        self.ops = ops
        # without a device (the host-logic tests run the cost models over the torch emulation of the C-ABI) the
        # "events" are host timestamps
        self.cuda = torch.cuda.is_available()
        self.peak_tf = PEAK_F32_TFLOPS if f32 else PEAK_F16X3_TFLOPS
        self.records = []   # (label, flops, bytes, e0, e1)
        self._saved = {}

    def _wrap(self, name, fn, cost):
        def wrapped(*a, **k):
            try:
                label, fl, by = cost(a, k)
            except Exception:  # a cost model must never break a measurement run
                label, fl, by = name, 0.0, 0.0
            if not self.cuda:
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    self.records.append((label, fl, by, t0, time.perf_counter()))
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            try:
                return fn(*a, **k)
            finally:
                e1.record(st)
                self.records.append((label, fl, by, e0, e1))
        return wrapped

    def __enter__(self):
        for name, cost in COSTS.items():
            fn = getattr(self.ops, name, None)
            if fn is None:
                continue
            self._saved[name] = name in self.ops.__dict__ and self.ops.__dict__[name]
            setattr(self.ops, name, self._wrap(name, fn, cost))
        return self

    def __exit__(self, *exc):
        for name, had in self._saved.items():
            if had:
                setattr(self.ops, name, had)
            else:
                delattr(self.ops, name)
        self._saved = {}
        return False

    def summary(self, steps=1, top=None):
        """One row per launch class, longest first: launches and ms per step, algorithmic GFLOP and MB per step, the
        bound, the achieved rate in that bound's unit and the fraction of its peak."""
        if self.cuda:
            torch.cuda.synchronize()
        agg = {}
        for label, fl, by, e0, e1 in self.records:
            r = agg.setdefault(label, [0, 0.0, 0.0, 0.0, 0.0])
            ms1 = e0.elapsed_time(e1) if self.cuda else (e1 - e0) * 1e3
            r[0] += 1
            r[1] += ms1
            r[4] = max(r[4], ms1)
            r[2] += fl
            r[3] += by
        rows = []
        for label, (n, ms, fl, by, mx) in agg.items():
            t_mfma = fl / (self.peak_tf * 1e12) * 1e3     # ms at the matrix-core peak
            t_hbm = by / (PEAK_HBM_TBS * 1e12) * 1e3      # ms at the HBM peak
            row = {'class': label, 'launches_per_step': round(n / steps, 2), 'ms_per_step': round(ms / steps, 4)}
            if n > steps:
                row['longest_launch_ms'] = round(mx, 4)
            if max(t_mfma, t_hbm) / max(n, 1) < 5e-3:
                row['bound'] = 'latency'
            elif t_mfma >= t_hbm:
                row.update(bound='mfma', achieved=round(fl / (ms * 1e-3) / 1e12, 1), unit='TFLOP/s-eq',
                           frac=round(t_mfma / ms, 4))
            else:
                row.update(bound='hbm', achieved=round(by / (ms * 1e-3) / 1e12, 3), unit='TB/s', frac=round(t_hbm / ms, 4))
            if fl:
                row['gflop_per_step'] = round(fl / steps / 1e9, 1)
            if by:
                row['mb_per_step'] = round(by / steps / 1e6, 1)
            rows.append(row)
        rows.sort(key=lambda r: -r['ms_per_step'])
        total = sum(r['ms_per_step'] for r in rows)
        if top is not None and len(rows) > top:
            rest = rows[top:]
            rows = rows[:top] + [{'class': 'other (%d classes)' % len(rest), 'bound': 'latency',
                                  'launches_per_step': round(sum(r['launches_per_step'] for r in rest), 2),
                                  'ms_per_step': round(sum(r['ms_per_step'] for r in rest), 4)}]
        return {'device_ms_per_step': round(total, 3), 'classes': rows}
