#!/usr/bin/env python
"""Throughput benchmark of the mmMOT per-frame-pair network forward on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frame-pairs/sec of the full eval-mode
``TrackingNet.forward`` at N_det=64, inputs resident in HBM.  Workload = the
N_det=64 configuration the metric is quoted on, ``configs[2]`` (= SURVEY cfg3):
Fusion C, N=M=64 (128 crops of 128x128), 2048 LiDAR points per detection.
One step = one pass of the hot path over one batch of ``--pairs`` synthetic
frame pairs per GPU (default 16 = 2 048 crops: 11.6 GB of the 288 GB; the ~1 ms of
small-kernel time per step does not grow with the batch).

Multi-GPU: one process per GPU (torch.distributed, RCCL), samples sharded, no
data-path collective, one flat result gather per step (weak scaling: per-GPU
work is fixed).  ``python bench.py --gpus N`` launched WITHOUT a distributed
environment re-executes itself under ``torch.distributed.run --nproc-per-node N``
(127.0.0.1 rendezvous, a free port); launched by torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE from the environment.  ``--dry`` runs the same
launcher / shard / gather / timing / JSON code on CPU tensors over gloo with a
stub step (no kernel runs; covered by tests/test_dist_cpu.py).

Arithmetic legs.  The headline (``value``, ``dtype``, ``roofline``) is the
``--trunk`` mode, by default ``f16x3`` - the fp32-class arithmetic (3-term fp16
hi/lo split, 22 significand bits, score error ~3e-5 against the fp32 reference).
The same run then times the other trunk mode(s) of ``--extra-trunks`` with the
same warm-up / step counts and reports them under ``extra`` (``f16q8``: both
correction terms on the fp8 matrix cores, score error ~3e-4 of the 1e-3 budget).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     - the dominant kernel (VGG trunk implicit GEMM), measured with HIP events around every
                 trunk launch inside the timed region; traffic from the committed PMC passes
  cpu_baseline - the oracle (CPU restatement of the reference) on the host cores
  parity       - L-inf of pair 0 against the committed output of the IMPORTED reference (tests/golden/f_*.npz,
                 same seed) and of the CPU-baseline pairs against the oracle
  end_to_end   - F_ref / F_exec per pair and the whole-step fraction of the f16x3 ceiling (north_star: >= 0.50)
  extra        - the other arithmetic legs; ``workloads``: the other BASELINE.json configs at their batch sizes
                 (cfg2 B=32, cfg4 32 pairs/GPU, cfg5 image-only 8 / LiDAR-only 32 pairs), each with value, roofline,
                 parity against its reference golden, clock / power telemetry; ``kernels``: per launch class of every
                 leg - ms per step, bound (mfma | hbm | latency), achieved TFLOP/s-equivalent or TB/s, fraction of that
                 peak - from HIP events around every operator call of two extra untimed steps (mmmot_amd/profiler.py);
                 ``prep``: point-cloud gather sweeps/s and crop-resize detections/s; ``rccl_world1``: the N-GPU step's
                 collective sequence on a one-rank nccl group (child process); latency of one reference-shaped call
  host / per_rank - NUMA node and cores the rank bound itself to (sysfs of its GPU), mean shader clock and package power
                 over the timed steps (hwmon); inputs: three synthetic input sets rotate over the steps (``--input-sets``)
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mmmot_amd.dist import gather_results, shard_range  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402

METRIC = 'frame-pairs/sec (fusion+affinity fwd) at N_det=64; affinity L∞ vs CPU ref'
WORKLOADS = {
    # name: (fusion, affinity_op, softmax_mode, N, M, S, pts/det, reference golden of the FIRST pair of the batch)
    'cfg3': ('C', 'multiply', 'none', 64, 64, 128, 2048, 'f_cfg3_C'),
    'cfg2': ('A', 'multiply', 'none', 32, 32, 64, 512, 's4_cfg2_A'),
    'cfg4': ('C', 'minus_abs', 'dual_add', 128, 128, 64, 512, 'f_cfg4_C'),
    'tiny': ('C', 'multiply', 'none', 6, 5, 32, 40, None),
}
# seed of the first pair of a workload's batch = the seed its reference golden was generated with (oracle/gen_golden.py)
SEED0 = {'cfg2': 1004}
# BASELINE.json `configs` beside the headline one, timed in the same default run and reported under extra.workloads:
# (key, workload, pairs per step per GPU, modality rows).  cfg2 = configs[1] (batch = 32 frame pairs), cfg4 = configs[3]
# (256 pairs over 8 GPUs = 32 per GPU), cfg5 = configs[4] (image-only / LiDAR-only rows at N_det = 64, cfg3 shapes).
EXTRA_WORKLOADS = [
    ('cfg2_b32', 'cfg2', 32, (0, 1, 2)),
    ('cfg4_b32_per_gpu', 'cfg4', 32, (0, 1, 2)),
    ('cfg5_image_only', 'cfg3', 8, (0,)),
    ('cfg5_lidar_only', 'cfg3', 32, (1,)),  # 8.4 M points per step: the ~1 ms of small launches is amortised (8: -10 %)
]
PROFILE_STEPS = 2  # untimed steps behind every leg's timed region that run under the launch profiler (extra.kernels)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: BF16/F16 MFMA dense peak (AMD's 5 PF figure is 2:1 sparse)
HEADLINE_TRUNK = 'f16x3'       # fp32-class arithmetic: what `value` is measured in unless --trunk says otherwise
BASE_KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True,
               appear_fpn=False, point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2',
               end_mode='avg', test_mode=2, neg_threshold=0.2, dropblock=0, use_dropout=False)


def reference_flops_per_pair(N, M, S, P, fusion):
    """F_ref of SURVEY 8d (reference-as-written FLOPs, 2 x MAC)."""
    L = N + M
    f_fus = 524288 if fusion in ('A', 'B') else 1048576
    return 2 * (L * (305856 * S * S + 204800) + P * 991625 + L * 262144 + 2.36e6 + L * f_fus + 3 * L * 393472 +
                3 * N * M * 852096 + 3 * L * 327808)


def executed_flops_per_pair(N, M, S, P, fusion, rows=(0, 1, 2), pn_gram=True):
    """F_exec: FLOPs (2 x MAC) of the math this build actually executes per frame pair.  Against
    F_ref: STN trunks removed (-282 816 MAC/pt), 1088->512 split (-524 288 MAC/pt, +524 288 MAC/det); added: the Gram
    matrices behind the statistics of conv5 (128 x 128: +16 384 MAC/pt) and - round 5 - of PointNet_v1.conv1 (64 x 64:
    +4 096 MAC/pt; it replaced a second 64->512 GEMM pass of 32 768 MAC/pt, so F_exec per point FELL by 28 672 MAC: a
    faster step now shows as more pairs/s at an unchanged fraction).  Each GEMM of the path is counted once, in the pass
    that normalises and reduces it.  The conv trunk term is F_ref's.  Single-modality rows drop the other branch and
    the fusion module, and run the head on one row.  ``pn_gram=False``: the legs that keep the second 64->512 GEMM pass
    instead of the 64 x 64 Gram matrix (exact-fp32 mode, MMMOT_PN_GRAM=0)."""
    L = N + M
    nR = len(rows)
    img = (0 in rows) or (2 in rows)
    pts = (1 in rows) or (2 in rows)
    f = 0.0
    if img:
        f += L * (305856 * S * S + 204800)
    if pts:
        f += P * (184521 + (4096 if pn_gram else 32768) + 16384) + L * (524288 + 262144)
    if nR == 3:
        f += L * (524288 if fusion in ('A', 'B') else 1048576)
    f += nR * L * 393472 + nR * N * M * 852096 + nR * L * 327808
    return 2.0 * f


# ---- the ONE stdout line -------------------------------------------------------------------------------------------
# The driver keeps a bounded tail of stdout and parses its last line: round 5's 20.8 KB line was cut and the round went
# unmeasured.  So: the full record (every leg, per-launch-class tables, telemetry prose) goes to
# gpurun_out/bench_detail_n<N>.json and to ONE stderr line prefixed ``BENCH_DETAIL `` (``--detail stdout`` prints it as
# an EARLIER stdout line instead); stdout carries exactly one JSON line of at most LINE_LIMIT characters, built by
# compact_line() below (tests/test_bench_line.py: size, round trip, required keys on a real round-5 record).
LINE_LIMIT = 6000
REQUIRED_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'end_to_end', 'parity')
ROOFLINE_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch',
                 'avg_launch_ms', 'trunk_share_of_step', 'share_of_step')


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 2] + '..'


def _sig(x, d=4):
    return round(x, d) if isinstance(x, float) else x


def compact_line(out, limit=LINE_LIMIT):
    """The final stdout line from the full record ``out``: contract fields, roofline, cpu_baseline, end_to_end, parity,
    one row per other BASELINE config, the launch classes that cost most / sit furthest below their roofline.
    Optional blocks are dropped (least important first) until the line fits ``limit``; the contract fields never are -
    if they alone do not fit, that is a bug and this raises."""
    c = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                 'scaling', 'vs_baseline', 'dtype', 'data')}
    cfg = out.get('config') or {}
    c['config'] = {k: (_short(v, 150) if isinstance(v, str) else v) for k, v in cfg.items()}
    rf = out.get('roofline') or {}
    c['roofline'] = {k: (_short(rf[k], 90) if isinstance(rf[k], str) else rf[k]) for k in ROOFLINE_KEYS if k in rf}
    cb = out.get('cpu_baseline')
    c['cpu_baseline'] = None if cb is None else {
        k: (_short(cb[k], 150) if isinstance(cb[k], str) else cb[k])
        for k in ('value', 'unit', 'cores', 'threads', 'host_cores', 'kind', 'sample') if k in cb}
    ee = out.get('end_to_end') or {}
    c['end_to_end'] = {'ref_gflop_per_pair': ee.get('ref_gflop_per_pair'), 'exec_gflop_per_pair': ee.get('exec_gflop_per_pair'),
                       'exec_tflops_equiv': ee.get('exec_tflops_equiv'),
                       'whole_step_frac': ee.get('whole_step_frac_of_f16x3_peak')}
    par = out.get('parity') or {}
    c['parity'] = {k: (_sig(par[k], 8)) for k in ('tolerance', 'linf_vs_reference_golden', 'linf_vs_cpu_oracle') if k in par}
    ex = out.get('extra') or {}
    optional = []   # (key, value) in DROP order: the first entries go first when the line is too long
    host = out.get('host') or {}
    tele = host.get('telemetry') or {}
    c['host'] = {'numa_node': host.get('numa_node'), 'cpus': host.get('cpus'), 'bound': host.get('bound'),
                 'sclk_mhz': tele.get('sclk_mhz'), 'power_w': tele.get('power_w')}
    if 'gather_ok' in out:
        c['gather_ok'] = out['gather_ok']
    if out.get('per_rank'):
        c['per_rank'] = out['per_rank']
    if out.get('solo'):
        c['solo'] = out['solo']
    wls = ex.get('workloads') or {}
    if wls:
        c['workloads'] = {'columns': ['pairs/s', 'whole_step_frac', 'ms_per_step', 'roofline_frac', 'linf_vs_reference'],
                          **{k: [w.get('value'), w.get('whole_step_frac_of_f16x3_peak'), w.get('ms_per_step'),
                                 (w.get('roofline') or {}).get('frac'), _sig(w.get('linf_vs_reference_golden'), 8)]
                             for k, w in wls.items()}}
    arith = {}
    for k, leg in ex.items():
        if isinstance(leg, dict) and 'value' in leg and 'dtype' in leg:
            arith[k] = [leg.get('value'), (leg.get('roofline') or {}).get('frac'), _sig(leg.get('linf_vs_reference_golden'), 8)]
    if arith:
        c['arithmetic_legs'] = {'columns': ['pairs/s', 'roofline_frac', 'linf_vs_reference'], **arith}
    lat = ex.get('latency') or {}
    if lat:
        c['latency_ms_b1'] = {'eager': lat.get('latency_ms_b1'), 'plan_cached': lat.get('latency_ms_b1_plan_cached'),
                              'hipgraph': lat.get('latency_ms_b1_hipgraph_replay'), 'device': lat.get('device_ms_b1_eager')}
    pipe = ex.get('pipeline') or {}
    if pipe:
        c['pipeline'] = {k: pipe[k] for k in ('frames_per_s', 'frames_per_s_serial', 'ms_per_frame', 'stage_ms', 'bitwise_equal')
                         if k in pipe}
    rc = ex.get('rccl_world1') or {}
    if rc:
        c['rccl_world1'] = {'ok': rc.get('ok'), 'gather_us_per_step': rc.get('gather_us_per_step')}
    prep = ex.get('prep') or {}
    if prep:
        c['prep'] = {k: [v.get('value'), v.get('unit')] for k, v in prep.items() if isinstance(v, dict)}
    # launch classes: per leg the device time and the rows that cost >= 1.5 % of the step, [class, ms/step, bound, frac];
    # the other legs list what the headline leg does not (their trunk rows repeat the headline's fractions)
    kern = ex.get('kernels') or {}
    ktab = {}
    for leg, tab in kern.items():
        if not isinstance(tab, dict) or 'rows' not in tab:
            continue
        tot = tab.get('device_ms_per_step') or 0.0
        head_leg = leg.startswith('headline') or not any(not r[0].startswith('trunk') and r[3] != 'latency' and tot
                                                         and r[2] >= 0.015 * tot for r in tab['rows'])
        rows = [[_short(r[0], 44), _sig(r[2], 3), r[3], _sig(r[5], 3)] for r in tab['rows']
                if r[3] != 'latency' and tot and r[2] >= 0.015 * tot and (head_leg or not r[0].startswith('trunk'))]
        ktab[leg] = {'device_ms_per_step': tot, 'rows': rows[:8 if leg.startswith('headline') else 6]}
    if ktab:
        c['kernels'] = dict(ktab, columns=['class', 'ms/step', 'bound', 'frac of that peak'])
    c['detail'] = out.get('detail_file')

    def line():
        return json.dumps(c, separators=(',', ':'))

    # drop order when over the limit: kernel rows of the non-headline legs (longest tables first), then the headline
    # table, then the small optional blocks
    drops = [('kernels', leg) for leg in list(ktab)[::-1]] + [('prep',), ('rccl_world1',), ('latency_ms_b1',), ('pipeline',),
                                                                 ('arithmetic_legs',), ('solo',), ('host',), ('workloads',)]
    dropped = []
    for d in drops:
        if len(line()) <= limit:
            break
        if len(d) == 2:
            if d[1] in c.get('kernels', {}):
                del c['kernels'][d[1]]
                dropped.append('kernels.' + d[1])
                if set(c['kernels']) <= {'columns'}:
                    del c['kernels']
        elif d[0] in c:
            del c[d[0]]
            dropped.append(d[0])
        if dropped:
            c['dropped_for_size'] = dropped
    s = line()
    if len(s) > limit:
        raise RuntimeError('bench line is %d characters (> %d) even without the optional blocks' % (len(s), limit))
    missing = [k for k in REQUIRED_KEYS if k not in c]
    if missing:
        raise RuntimeError('bench line misses %s' % missing)
    return s


# stdout belongs to the ONE line: libraries write to the process's stdout too - RCCL prints a version banner through C stdio
# (block-buffered when stdout is a file or a pipe, so it comes out at EXIT, behind the JSON line: measured on an MI355X with
# --force-dist), gloo prints its rank lines - and the driver parses the LAST line.  claim_stdout() therefore points file
# descriptor 1 at stderr for the rest of the process (Python's and every library's output follows it there) and keeps a
# private duplicate of the real stdout that only emit_line() writes to.
_REAL_STDOUT = None


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    data = (text + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(text + '\n')
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def emit(out, world, detail='stderr'):
    """rank 0: the full record to gpurun_out/bench_detail_n<world>.json and a BENCH_DETAIL line, then the ONE stdout line"""
    name = 'gpurun_out/bench_detail_n%d.json' % world
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, name), 'w') as f:
            json.dump(out, f, indent=1)
        out['detail_file'] = name
    except OSError:
        out['detail_file'] = None
    full = 'BENCH_DETAIL ' + json.dumps(out)
    if detail == 'stdout':
        emit_line(full)
    elif detail == 'stderr':
        print(full, file=sys.stderr, flush=True)
    emit_line(compact_line(out))


def dry_record(args, world, G, dt, per_rank, gather_ok, place):
    """--dry: a record with EVERY key and table the GPU run produces (placeholder numbers, the real strings and the
    profiler's real launch-class names at 25 rows per leg), so that the size / key checks of the stdout line run on
    the real shape without a GPU (tests/test_bench_line.py, tests/test_dist_cpu.py)."""
    from mmmot_amd.profiler import COSTS  # noqa: F401 - the label strings below are the profiler's longest ones
    fusion, aff, sm, N, M, S, pts, gold = WORKLOADS[args.workload]
    B = per_rank['pairs'][0]
    names = ['trunk conv1_1+conv1_2 (fused) 3->64->64 @128x128 +pool', 'trunk conv 256->256 @32x32 +pool',
             'A-resident GEMM 128->1024 (norm+relu prologue, column sums only)', 'rows GEMM 512->1024 (pair prologue, reduced)',
             'Gram matrix of the 128-channel rows (conv5 statistics)', 'GroupNorm scale/shift from the Gram matrix',
             'segment mean, 1024 channels (norm+relu rows)', 'SkipPool head (LN, 1x1, LN, 1x1, LN)']
    table = {'device_ms_per_step': 54.068,
             'rows': [[names[i % len(names)] + ' #%d' % i, 1.0, round(7.0 / (i + 1), 4), ('mfma', 'hbm', 'latency')[i % 3],
                       None if i % 3 == 2 else 443.7, None if i % 3 == 2 else 0.5325] for i in range(25)]}
    roof = {'bound': 'mfma', 'kernel': 'conv3x3_hl16_patch_kernel (VGG16-BN trunk layers 2-13, 12 launches/step)',
            'achieved': 0.0, 'peak': 833.3, 'unit': 'TFLOP/s', 'frac': 0.0, 'traffic': 4034477625,
            'traffic_unit': 'bytes/launch (mean)', 'traffic_source': 'profiles/r05/rocprofv3_pmc_FETCH_SIZE_cfg3_pairs1_f16x3.txt '
            '(KB per dispatch summed over the kernel\'s launches of 3 steps)', 'algorithmic_bytes_per_launch': 3276460032,
            'peak_basis': '2500 TFLOP/s dense f16 MFMA / 3 MFMAs per algorithmic product (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, '
            'fp32 accumulate)', 'avg_launch_ms': 0.0, 'trunk_share_of_step': 0.0,
            'flops_basis': 'algorithmic 2*9*Cin*Cout per output pixel (conv1_1 counted at Cin=3)'}
    tele = {'sclk_mhz': None, 'power_w': None, 'samples': 0}

    def leg(key):
        return {'value': 0.0, 'ms_per_step': 0.0, 'dtype': 'f16x3', 'roofline': dict(roof), 'exec_gflop_per_pair': 1412.3,
                'exec_tflops_equiv_per_gpu': 0.0, 'whole_step_frac_of_f16x3_peak': 0.0,
                'inputs': '3 sets x 16 pairs rotated over the steps, 1350 MiB resident', 'telemetry': dict(tele),
                'linf_vs_reference_golden': 2.4e-05, 'unit': 'frame-pairs/s',
                'config': '%s: Fusion C, N=M=128, 128x128 crops, 2048 pts/det, 32 pairs/step/GPU, modality rows (0, 1, 2)' % key}
    out = {
        'metric': METRIC, 'value': None, 'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': None, 'data': 'dry-run stub (no kernel executed)', 'gather_ok': bool(gather_ok),
        'config': {'workload': '%s: Fusion %s, %s/%s, N=M=%d (%d crops of %dx%d), %d pts/det; %d pairs/step/GPU' % (
            args.workload, fusion, aff, sm, N, N + M, S, S, pts, B), 'pairs_per_step_per_gpu': B, 'global_pairs': G,
            'trunk': None, 'parallelism': 'sample-sharded x%d, flat all_gather of scores (gloo, CPU)' % world,
            'hipgraph': False, 'rccl': False},
        'roofline': roof,
        'cpu_baseline': {'value': None, 'unit': 'frame-pairs/s', 'cores': None, 'threads': None, 'host_cores': os.cpu_count(),
                         'kind': 'port', 'sample': 'dry run: not timed. ' + 'x' * 300},
        'end_to_end': {'ref_gflop_per_pair': 1824.6, 'ref_tflops_equiv': 0.0, 'exec_gflop_per_pair': 1412.3,
                       'exec_tflops_equiv': 0.0, 'whole_step_frac_of_f16x3_peak': 0.0, 'basis': 'x' * 330},
        'parity': {'tolerance': 1e-3, 'linf_vs_reference_golden': None, 'linf_vs_cpu_oracle': None,
                   'golden': 'tests/golden/%s.npz' % gold},
        'inputs': 'stub', 'host': dict(place, telemetry=tele, note='x' * 200), 'per_rank': per_rank,
        'solo': {'ms_per_step': 0.0, 'pairs': B, 'steps': args.steps, 'sclk_mhz': None, 'power_w': None} if world > 1 else None,
        'extra': {'f16q8': leg('cfg3'), 'f16x3_hipgraph': {'value': 0.0, 'ms_per_step': 0.0, 'dtype': 'f16x3',
                                                           'linf_vs_reference_golden': 2.4e-05},
                  'workloads': {k: leg(k) for k, _, _, _ in EXTRA_WORKLOADS},
                  'rccl_world1': {'backend': 'nccl (RCCL)', 'world_size': 1, 'gather_us_per_step': 0.0,
                                  'max_over_ranks_ok': True, 'ok': True},
                  'latency': {'latency_ms_b1_hipgraph_replay': 0.0, 'device_ms_b1_eager': 0.0, 'shape': 'x' * 90,
                              'latency_ms_b1': 0.0, 'latency_ms_b1_plan_cached': 0.0, 'includes': 'x' * 200},
                  'pipeline': {'frames_per_s': 0.0, 'frames_per_s_serial': 0.0, 'ms_per_frame': 0.0, 'bitwise_equal': True,
                               'stage_ms': {'h2d': 0.0, 'prep_points': 0.0, 'crop_resize': 0.0, 'forward': 0.0, 'scores_d2h': 0.0},
                               'workload': 'x' * 200},
                  'kernels': dict({k: dict(table) for k in ['headline_%s_f16x3' % args.workload] +
                                   [k for k, _, _, _ in EXTRA_WORKLOADS]}, columns=['x'] * 6, how='x' * 500),
                  'prep': {k: {'value': 0.0, 'unit': 'detections/s', 'workload': 'x' * 120}
                           for k in ('point_gather', 'crop_resize_normalize_fp32', 'crop_resize_u8')}}}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='cfg3', choices=sorted(WORKLOADS))
    ap.add_argument('--pairs', type=int, default=16, help='frame pairs per step per GPU')
    ap.add_argument('--cpu-pairs', type=int, default=3, help='timed pairs of the CPU baseline (0 disables)')
    ap.add_argument('--no-gather', action='store_true')
    ap.add_argument('--graph', action='store_true', help='capture the launch sequence of one step in a hipGraph and replay it')
    ap.add_argument('--trunk', default=HEADLINE_TRUNK, choices=['f16q8', 'f16x3', 'f32'],
                    help="headline VGG trunk arithmetic: f16x3 = fp16 matrix cores, 3-term hi/lo split (fp32-class); "
                         "f16q8 = fp16 main term + fp8 correction terms; f32 = exact fp32 MFMA")
    ap.add_argument('--extra-trunks', default=None,
                    help="comma list of further trunk modes timed in the same run and reported under 'extra' "
                         "(default: the other one of f16x3 / f16q8; 'none' disables)")
    ap.add_argument('--no-latency', action='store_true', help='skip the B=1 reference-call latency extra')
    ap.add_argument('--no-workloads', action='store_true',
                    help='skip extra.workloads (the other BASELINE.json configs: cfg2 B=32, cfg4 32 pairs/GPU, cfg5 rows)')
    ap.add_argument('--input-sets', type=int, default=3,
                    help='distinct synthetic input sets rotated over the steps (the last timed step runs set 0, the one the '
                         'parity checks look at): 3 x (100 MB crops + 50 MB points) at the default batch exceeds the 256 MiB '
                         'Infinity Cache, so no step finds its inputs cached by the step before')
    ap.add_argument('--no-profile', action='store_true',
                    help='skip extra.kernels (HIP events around every operator call of a few extra, untimed steps) and '
                         'extra.prep (point-cloud gather / crop-resize throughput)')
    ap.add_argument('--no-bind', action='store_true', help='leave the CPU affinity of the rank alone (default: the cores '
                    'of the NUMA node its GPU hangs off)')
    ap.add_argument('--rows', default='0,1,2', help='modality rows of the headline workload (cfg5: 0 = image-only, 1 = LiDAR-only)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise torch.distributed (RCCL) and run the collective path of the N-GPU run even with one GPU')
    ap.add_argument('--latency-only', action='store_true', help='run only the B=1 latency leg (profiling)')
    ap.add_argument('--rccl-world1-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--device', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--global-pairs', type=int, default=None,
                    help='size of the GLOBAL batch (default: gpus x --pairs); not a multiple of --gpus = uneven shards '
                         '(first ranks one pair more), gathered over the ragged form of the result gather')
    ap.add_argument('--detail', default='stderr', choices=['stderr', 'stdout', 'none'],
                    help='where the full record goes as ONE line prefixed "BENCH_DETAIL " (always also written to '
                         'gpurun_out/bench_detail_n<N>.json); stdout carries the compact line last either way')
    ap.add_argument('--dry', action='store_true',
                    help='CPU / gloo dry run of the launcher, sharding, gather, timing and JSON with a stub step')
    return ap.parse_args(argv)


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a distributed environment: become N ranks on this node."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)



# ---- host placement and device telemetry of a rank ---------------------------------------
def gpu_sysfs_dir(index):
    """sysfs directory of cuda:<index> (/sys/bus/pci/devices/<domain:bus:dev.fn>), or None."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = '%04x:%02x:%02x.0' % (int(getattr(p, 'pci_domain_id', 0)), int(p.pci_bus_id), int(p.pci_device_id))
        d = os.path.join('/sys/bus/pci/devices', bdf)
        return d if os.path.isdir(d) else None
    except Exception:
        return None


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(text):
    cpus = set()
    for part in (text or '').split(','):
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        elif part.strip():
            cpus.add(int(part))
    return cpus


def bind_to_gpu_numa_node(index, enable=True):
    """Pin this rank (and every thread it starts later: the launch thread, torch's intra-op pool) to the cores of the
    NUMA node its GPU is attached to, so that eight ranks on one node neither share cores nor launch across sockets.
    Returns what was done: {'numa_node', 'cpus' (count), 'bound'}; a missing sysfs entry or node -1 leaves the
    affinity alone."""
    info = {'numa_node': None, 'cpus': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
            'bound': False}
    d = gpu_sysfs_dir(index) if index is not None else None
    if d is None:
        return info
    node = _read(os.path.join(d, 'numa_node'))
    cpus = _cpulist(_read(os.path.join(d, 'local_cpulist')))
    info['numa_node'] = int(node) if node not in (None, '') else None
    if enable and cpus and info['numa_node'] is not None and info['numa_node'] >= 0 and hasattr(os, 'sched_setaffinity'):
        allowed = cpus & os.sched_getaffinity(0)
        if allowed:
            os.sched_setaffinity(0, allowed)
            info.update(cpus=len(allowed), bound=True)
    return info


class Telemetry:
    """Sustained shader clock and package power of one GPU while a leg's timed steps run: a host thread samples the
    device's hwmon files (freq1_input = sclk in Hz, power1_average / power1_input in uW) every `period` seconds."""

    def __init__(self, index, period=0.25):
        import glob
        self.period = period
        d = gpu_sysfs_dir(index)
        hw = sorted(glob.glob(os.path.join(d, 'hwmon', 'hwmon*'))) if d else []
        self.f_clk = self.f_pow = None
        for h in hw:
            if self.f_clk is None and os.path.exists(os.path.join(h, 'freq1_input')):
                self.f_clk = os.path.join(h, 'freq1_input')
            for name in ('power1_average', 'power1_input'):
                if self.f_pow is None and os.path.exists(os.path.join(h, name)):
                    self.f_pow = os.path.join(h, name)
        self.clk, self.pow = [], []
        self._stop = None
        self._thr = None

    def _run(self):
        while not self._stop.is_set():
            c, w = _read(self.f_clk) if self.f_clk else None, _read(self.f_pow) if self.f_pow else None
            try:
                if c:
                    self.clk.append(float(c) / 1e6)
                if w:
                    self.pow.append(float(w) / 1e6)
            except ValueError:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self.clk, self.pow = [], []
        if self.f_clk or self.f_pow:
            self._stop = threading.Event()
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join()
            self._thr = None
        return False

    def result(self):
        mean = lambda v: round(sum(v) / len(v), 1) if v else None
        return {'sclk_mhz': mean(self.clk), 'power_w': mean(self.pow), 'samples': max(len(self.clk), len(self.pow))}


def stub_results(pairs, N, M, seed0):
    """--dry: reference-shaped per-sample results from a seed (no kernel, no oracle)."""
    out = []
    for i in range(pairs):
        g = torch.Generator().manual_seed(seed0 + i)
        L = N + M
        out.append((torch.rand(3, L, generator=g), [torch.rand(3, N, M, generator=g)], torch.rand(3, L, generator=g),
                    torch.rand(3, L, generator=g)))
    return out


def time_steps(step, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(steps):
        res = step()
    barrier()
    return time.perf_counter() - t0, res


def max_over_ranks(dt, world, dev):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def roofline_of(trunk, events, eng, B, workload, dt, rows=(0, 1, 2)):
    """Roofline entry of the dominant kernel from the HIP events recorded around every trunk launch."""
    per_layer = {}
    tagged = [ev for ev in events if isinstance(ev[0], str)]   # non-trunk launches the engine brackets ('pn5')
    events = [ev for ev in events if not isinstance(ev[0], str)]
    if not events:
        # LiDAR-only rows: no trunk.  Dominant kernel = PointNet conv5 128 -> 1024 (normalise + segment-sum pass).
        ms = sum(e0.elapsed_time(e1) for _, _, _, _, e0, e1 in tagged)
        fl = sum(2.0 * r * k * n for _, r, k, n, _, _ in tagged)
        f16 = trunk != 'f32'
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if f16 else PEAK_F32_MFMA_TFLOPS
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_src = None, None
        try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (profiles/traffic.json)
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
                tr = json.load(f).get('%s_lidar/%s' % (workload, trunk))
            if tr:
                traffic = round((tr['fetch_bytes_per_pair'] + tr['write_bytes_per_pair']) * B / tr['launches_per_step'])
                traffic_src = tr['source']
        except (OSError, ValueError, KeyError):
            pass
        roof = {'bound': 'mfma', 'kernel': 'gemm_wreg128_kernel (PointNet conv5 128->1024 + GroupNorm + ReLU + per-detection mean, '
                'consumer pass with the weights in registers; 1 launch/step)', 'achieved': round(ach, 2), 'peak': round(peak, 1),
                'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_unit': 'bytes/launch (mean)',
                'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': round(sum(r * k * 4.0 for _, r, k, _, _, _ in tagged) / max(len(tagged), 1)),
                'avg_launch_ms': round(ms / max(len(tagged), 1), 4), 'share_of_step': round(ms / (dt * 1e3), 4),
                'flops_basis': 'algorithmic 2*K*N per point', 'peak_basis': '2500/3 (f16x3 row GEMMs)' if f16 else 'fp32 MFMA'}
        return roof, {}
    for li, rows, cin, cout, e0, e1 in events:
        ms = e0.elapsed_time(e1)
        a = per_layer.setdefault(li, [0.0, 0.0, 0, rows, cin, cout])
        a[0] += ms
        a[1] += 2.0 * rows * 9 * cin * cout
        a[2] += 1
    trunk_ms = sum(a[0] for a in per_layer.values())
    if trunk == 'f16x3':
        # dominant kernel = the hl16 trunk kernel (layers 1..12); conv1_1 (Cin=3) is computed in the first launch's prologue
        dom = {li: a for li, a in per_layer.items() if li != 0}
        kname = 'conv3x3_hl16_patch_kernel (VGG16-BN trunk layers 2-13, 12 launches/step)'
        peak = PEAK_F16_MFMA_TFLOPS / 3.0
        peak_basis = ('%.0f TFLOP/s dense f16 MFMA / 3 MFMAs per algorithmic product (a_hi*w_hi + a_hi*w_lo + '
                      'a_lo*w_hi, fp32 accumulate)' % PEAK_F16_MFMA_TFLOPS)
    elif trunk == 'f16q8':
        ql = getattr(eng, 'q8_layers', None)  # the layers that run the hq8 arithmetic (default conv3_1 .. conv5_3)
        dom = {li: a for li, a in per_layer.items() if li != 0 and (ql is None or li in ql)}
        kname = 'conv3x3_hl16_patch_kernel<Q8> (the %d VGG16-BN trunk launches per step that run the hq8 arithmetic)' % len(dom)
        peak = PEAK_F16_MFMA_TFLOPS / 2.0
        peak_basis = ('%.0f TFLOP/s dense f16 MFMA / 2 f16-MFMA equivalents per algorithmic product (a_hi*w_hi on the '
                      'f16 cores + both correction terms in one block-scaled fp8 K=64 MFMA at twice the f16 rate)'
                      % PEAK_F16_MFMA_TFLOPS)
    else:
        dom = per_layer
        kname = 'conv3x3_kernel (VGG16-BN trunk, 13 launches/step)'
        peak = PEAK_F32_MFMA_TFLOPS
        peak_basis = 'fp32-input MFMA v_mfma_f32_32x32x2_f32 dense peak'
    conv_ms = sum(a[0] for a in dom.values())
    conv_fl = sum(a[1] for a in dom.values())
    n_launch = sum(a[2] for a in dom.values())
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0  # 0 with --graph: replays record no events
    # HBM/fabric bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process,
    # so the figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/traffic.json).
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            tr = json.load(f).get('%s/%s' % (workload, trunk))
        if tr:
            traffic = round((tr['fetch_bytes_per_pair'] + tr['write_bytes_per_pair']) * B / tr['launches_per_step'])
            traffic_src = tr['source']
    except (OSError, ValueError, KeyError):
        pass
    # algorithmic bytes per launch: every layer reads its input and weights once and writes its output once (4 B/value)
    alg_bytes = sum((a[3] * a[4] + (a[3] // (4 if li in (1, 3, 6, 9, 12) else 1)) * a[5] + 9 * a[4] * a[5]) * 4.0
                    for li, a in dom.items()) / max(len(dom), 1)
    roof = {'bound': 'mfma', 'kernel': kname, 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_unit': 'bytes/launch (mean)',
            'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': round(alg_bytes), 'peak_basis': peak_basis,
            'avg_launch_ms': round(conv_ms / max(n_launch, 1), 4), 'trunk_share_of_step': round(trunk_ms / (dt * 1e3), 4),
            'flops_basis': 'algorithmic 2*9*Cin*Cout per output pixel (conv1_1 counted at Cin=3)'}
    layers = {str(li): dict(ms_per_launch=a[0] / a[2], tflops=a[1] / (a[0] * 1e-3) / 1e12 if a[0] else 0, rows=a[3],
                            cin=a[4], cout=a[5]) for li, a in sorted(per_layer.items())}
    return roof, layers


def golden_linf(res0, name, rows=(0, 1, 2)):
    """L-inf of one sample's (det, links, new, end) against the committed output of the imported reference
    (``rows``: the modality rows the sample holds - the single-modality paths are rows 0 / 1 of the same golden)."""
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    if not os.path.exists(path):
        return None
    g = np.load(path)
    r = list(rows)
    det, links, new, end = res0
    return float(max(np.abs(det.cpu().numpy() - g['det'][r]).max(), np.abs(links[0].cpu().numpy() - g['link0'][r]).max(),
                     np.abs(new.cpu().numpy() - g['new'][r]).max(), np.abs(end.cpu().numpy() - g['end'][r]).max()))


def init_single_rank_env():
    """rendezvous environment of a one-rank process group on this node (127.0.0.1, a free port)"""
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(port))
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('LOCAL_RANK', '0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def rccl_world1_leg(dev):
    """rccl_world1_child() in a child process (its own process group and RCCL banner lines stay out of this process's
    stdout, which carries exactly one JSON line; a crash inside RCCL cannot take the benchmark line with it)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('MASTER_ADDR', 'MASTER_PORT', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--rccl-world1-child', '--device', str(dev.index or 0)],
                           capture_output=True, text=True, timeout=180, env=env)
        for line in reversed(p.stdout.splitlines()):
            if line.startswith('RCCL_WORLD1 '):
                return json.loads(line[len('RCCL_WORLD1 '):])
        return {'ok': False, 'error': 'child rc=%d: %s' % (p.returncode, (p.stderr or p.stdout)[-300:])}
    except Exception as e:  # noqa: BLE001
        return {'ok': False, 'error': '%s: %s' % (type(e).__name__, str(e)[:300])}


def rccl_world1_child(dev):
    """The exact collective call sequence of the N-GPU step on ONE GPU: init_process_group('nccl') with world size 1,
    barrier, the flat all_gather_into_tensor of the scores (equal and ragged form), the max-over-ranks all_reduce.
    Proves the RCCL branch executes on this box; says nothing about scaling.  Never raises: a failure is reported."""
    import torch.distributed as dist
    from mmmot_amd.dist import gather_flat
    info = {'backend': 'nccl (RCCL)', 'world_size': 1}
    try:
        init_single_rank_env()
        dist.init_process_group('nccl', device_id=dev)
        res = [(torch.rand(3, 128, device=dev), [torch.rand(3, 64, 64, device=dev)], torch.rand(3, 128, device=dev),
                torch.rand(3, 128, device=dev)) for _ in range(8)]  # 8 cfg3-shaped score sets
        got = gather_results(res, same_layout=True, force=True)
        ok = len(got) == 8 and all(torch.equal(a[1][0], b[1][0]) and torch.equal(a[0], b[0]) for a, b in zip(got, res))
        flat = torch.arange(1000, dtype=torch.float32, device=dev)
        rag = gather_flat(flat, equal=False, force=True)
        ok = ok and len(rag) == 1 and torch.equal(rag[0], flat)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gather_results(res, same_layout=True, force=True)
        torch.cuda.synchronize()
        info['gather_us_per_step'] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
        info['max_over_ranks_ok'] = abs(max_over_ranks(1.25, 2, dev) - 1.25) < 1e-12  # the all_reduce(MAX) of the timing
        info['ok'] = bool(ok)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported, the benchmark line must still be printed
        info['ok'] = False
        info['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    return info


def cpu_baseline(model, ins, res, fusion, aff, sm, N, M, n_timed):
    """The oracle (CPU restatement of the reference) on the host cores: >= 3 timed pairs after one warm-up pair,
    batch-1 loop like the reference (eval_seq.py:153), median s/pair.  Thread count: all cores is NOT the fastest
    on a many-core host (256 threads: 71 s/pair on the GPU box vs ~5 s at 32-64), so the count is calibrated on a
    proxy with the workload's two heavy parts (VGG trunk on 8 crops + PointNet feature MLP on 32 768 points)."""
    from oracle import restatement as R  # checker / baseline only - never on the product path
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(fusion=fusion, affinity_op=aff, softmax_mode=sm, neg_threshold=BASE_KW['neg_threshold'],
               score_arch=BASE_KW['score_arch'])
    host = os.cpu_count()
    proxy_c = ins[0][0][:8]
    pp = ins[0][1]['points'][:, :32768].transpose(-1, -2)
    ps = torch.arange(0, pp.shape[-1] + 1, 2048)

    def proxy():
        x = proxy_c
        for s in range(4):
            x = R.vgg_stage(x, sd, s)
        R.pointnet(pp, ps, sd)

    best_thr, best_t, sweep = 1, float('inf'), {}
    for thr in sorted({min(host, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(thr)
        with torch.no_grad():
            proxy()
            t1 = time.perf_counter()
            proxy()
            t = time.perf_counter() - t1
        sweep[thr] = round(t, 3)
        if t < best_t:
            best_thr, best_t = thr, t
    torch.set_num_threads(best_thr)
    times, linf = [], 0.0
    B = len(ins)
    with torch.no_grad():
        for i in range(n_timed + 1):
            dets, info, _ = ins[i % B]
            t1 = time.perf_counter()
            o = R.tracking_forward(sd, cfg, dets, info['points'], info['points_split'], [N, M])
            if i > 0:
                times.append(time.perf_counter() - t1)
            if res is not None:
                det, links, new, end = res[i % B]
                linf = max(linf, (links[0].cpu() - o[1][0]).abs().max().item(), (det.cpu() - o[0]).abs().max().item(),
                           (new.cpu() - o[2]).abs().max().item(), (end.cpu() - o[3]).abs().max().item())
    times.sort()
    med = times[len(times) // 2]
    base = {'value': round(1.0 / med, 4), 'unit': 'frame-pairs/s', 'cores': best_thr, 'threads': best_thr,
            'host_cores': host, 'kind': 'port',
            'sample': '%d timed pairs of the same workload after 1 warm-up pair, median s/pair %.3f (all: %s), torch %s '
                      'CPU fp32, batch-1 loop like the reference; %d threads = fastest of the calibration sweep %s '
                      '(s per proxy: VGG trunk on 8 crops + PointNet on 32768 points) on a %d-core host' % (
                          len(times), med, ' '.join('%.2f' % t for t in times), torch.__version__, best_thr, sweep, host)}
    return base, linf



def prep_leg(dev):
    """Throughput of the two device steps in front of the forward, synthetic inputs resident:
    the batched point-cloud gather (64 sweeps x 120 000 points x 4 floats, 16 boxes + the image-frustum-like filter
    polygon per sweep -> det_info['points'] / points_split) and crop + resize + normalise (32 detections of a
    1242 x 375 frame -> 224 x 224 crops, fp32 NCHW and the 8-bit form)."""
    from mmmot_amd import crops as CR
    from mmmot_amd import points as PT
    out = {}
    rng = np.random.default_rng(7)
    NS, P, NB = 64, 120000, 16
    eye = np.eye(4, dtype=np.float32)
    pts, planes, counts = [], [], []
    for _ in range(NS):
        pts.append(np.stack([rng.uniform(0, 70, P), rng.uniform(-30, 30, P), rng.uniform(-2.5, 1.0, P),
                             rng.uniform(0, 1, P)], 1).astype(np.float32))
        b = np.concatenate([rng.uniform([5, -20, -1.9], [50, 20, -1.2], (NB, 3)),          # centre (bottom face)
                            rng.uniform([3.2, 1.3, 1.4], [4.8, 1.8, 2.0], (NB, 3)),          # l, h, w
                            rng.uniform(-3.1, 3.1, (NB, 1))], 1).astype(np.float32)
        planes.append(PT.rbbox_planes(b, eye, eye))
        counts.append(NB)
    view = PT.rbbox_planes(np.array([[33.5, 0.0, -3.0, 56.0, 3.9, 63.0, 0.0]], dtype=np.float32), eye, eye)
    allpts = torch.from_numpy(np.concatenate(pts)).to(dev)
    rows_off = np.arange(NS + 1) * P
    planes_all = np.concatenate(planes + [view] * NS)
    filt = [NS * NB + i for i in range(NS)]

    def gather():
        return PT.gather_points_batched(allpts, rows_off, planes_all, counts, filters=filt, pad_empty=True,
                                        drop_reflectivity=True)
    rows, split = gather()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        gather()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['point_gather'] = {'value': round(n * NS / dt, 1), 'unit': 'sweeps/s', 'ms_per_batch': round(dt / n * 1e3, 3),
                           'workload': '%d sweeps x %d points x 4 floats per batch, %d boxes + 1 filter polygon per sweep; '
                                       '%d points kept' % (NS, P, NB, int(split[-1])),
                           'hbm_tbs_algorithmic': round((2 * allpts.numel() * 4 + rows.numel() * 4) / (dt / n) / 1e12, 3),
                           'includes': 'table upload, count pass, the one split read-back, scatter pass'}
    H, W, ND, S = 375, 1242, 32, 224
    frame = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).to(dev)
    x1, y1 = rng.uniform(0, W - 200, ND), rng.uniform(0, H - 150, ND)
    bb = np.stack([x1, y1, x1 + rng.uniform(30, 190, ND), y1 + rng.uniform(30, 140, ND)], 1)
    for key, fn in (('crop_resize_normalize_fp32', lambda: CR.crop_resize_normalize(frame, bb, S)),
                    ('crop_resize_u8', lambda: CR.crop_resize_u8(frame, bb, S))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = {'value': round(n * ND / dt, 1), 'unit': 'detections/s', 'us_per_frame': round(dt / n * 1e6, 1),
                    'workload': '%d detections of a %dx%d frame -> %dx%d crops per call (one frame per call, like the '
                                'reference loop)' % (ND, W, H, S, S)}
    return out


def latency_b1(dev, trunk):
    """What eval_seq.py:143-153 actually calls: ONE frame pair per call through ``model(dets, det_info, dets_split)``
    at the shape of BASELINE.json configs[0] (Fusion A, N=10, M=12, 224x224 crops, ragged ~300 pts/det), inputs on the
    device, INCLUDING the per-call host work (points_split D2H, BatchPlan build + table uploads, launches) and a
    final synchronize.  Every call sees a different points_split (plan-cache miss), like consecutive frames."""
    from mmmot_amd import TrackingNet
    from mmmot_amd.weights import init_module
    model = TrackingNet(**dict(BASE_KW, score_fusion_arch='A', affinity_op='multiply', softmax_mode='none'))
    init_module(model, seed=0)
    model.eval().to(dev)
    model.set_trunk(trunk)
    ins = []
    for i in range(12):
        dets, info, ds = make_pair(10, 12, 224, 300, seed=3000 + i, ragged=True)
        ins.append((dets.to(dev), {k: v.to(dev) for k, v in info.items()}, ds))
    ts, ts_cached = [], []
    with torch.no_grad():
        for rep in range(2):
            for i, (dets, info, ds) in enumerate(ins):
                if rep == 0:
                    model._plans.clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = model(dets, info, ds)
                out[1][0].sum().item()  # the consumer reads the scores on the host
                (ts if rep == 0 else ts_cached).append(time.perf_counter() - t0)
    ts, ts_cached = sorted(ts[2:]), sorted(ts_cached)
    # fixed batch shape: hipGraph replay of the same launch sequence (inputs copied into the captured buffers)
    dets, info, ds = ins[0]
    pts = info['points'].reshape(-1, 3).contiguous()
    ps0 = info['points_split'].reshape(-1).long().cpu().numpy()
    plan = model.make_plan([([10, 12], ps0)], 224)
    graphed = model.capture(plan, dets, pts)
    tg = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = graphed(dets, pts)
        out[0][1][0].sum().item()
        tg.append(time.perf_counter() - t0)
    tg.sort()
    # device time of one forward: HIP events around an eager call with everything cached
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        torch.cuda.synchronize()
        e0.record()
        model(dets, info, ds)
        e1.record()
        torch.cuda.synchronize()
    return {'latency_ms_b1_hipgraph_replay': round(tg[len(tg) // 2] * 1e3, 3), 'device_ms_b1_eager': round(e0.elapsed_time(e1), 3),'shape': 'cfg1: Fusion A, N=10, M=12, 224x224 crops, ragged ~300 pts/det, B=1, trunk %s' % trunk,
            'latency_ms_b1': round(ts[len(ts) // 2] * 1e3, 3), 'latency_ms_b1_plan_cached': round(ts_cached[len(ts_cached) // 2] * 1e3, 3),
            'includes': 'points_split D2H + BatchPlan build/upload (plan-cache miss) + launches + D2H of the scores; the module '
                        'call launches the trunk first and PointNet beside it (MMMOT_IMAGE_FIRST / MMMOT_PN_BESIDE_TRUNK)'}


def pipeline_leg(dev, trunk, n_frames=100):
    """The rows of SURVEY section 8 chained per frame the way eval_seq.py:140-160 + dataset/test_seq_dataset.py:176-246 +
    tracking_model.py:68-83 chain them, on a synthetic KITTI-shaped sequence (1242 x 375 frames, ~120 k-point sweeps,
    10-12 detections per frame = BASELINE configs[0]'s shape): per frame H2D -> prep_points -> crop_resize_u8, per pair
    TrackingNet.forward -> scores_for_solver; frame t+1's upload and preparation on a side stream under pair t's forward
    (mmmot_amd/pipeline.py).  Reports frames/s of the overlapped and the serial order, the serial order's device ms per
    stage, and that both orders return bitwise equal scores."""
    from concurrent.futures import ThreadPoolExecutor
    from mmmot_amd import TrackingNet
    from mmmot_amd.pipeline import FrameFeed, stage_times, time_sequence
    from mmmot_amd.synth import make_frame
    from mmmot_amd.weights import init_module
    model = TrackingNet(**dict(BASE_KW, score_fusion_arch='A', affinity_op='multiply', softmax_mode='none'))
    init_module(model, seed=0)
    model.eval().to(dev)
    model.set_trunk(trunk)
    rng = np.random.default_rng(5)
    ndet = rng.integers(10, 13, n_frames)
    with ThreadPoolExecutor(max(1, min(16, len(os.sched_getaffinity(0))))) as pool:
        feeds = [FrameFeed(*f) for f in pool.map(lambda t: make_frame(7000 + t, 120000, int(ndet[t])), range(n_frames))]
    fps_o, dt_o, res_o = time_sequence(model, feeds, 224, overlap=True)
    fps_s, dt_s, res_s = time_sequence(model, feeds, 224, overlap=False)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
               for a, b in zip(res_o, res_s))
    st = stage_times(model, feeds[:40], 224)
    return {'frames_per_s': round(fps_o, 1), 'frames_per_s_serial': round(fps_s, 1), 'ms_per_frame': round(1e3 / fps_o, 3),
            'ms_per_frame_serial': round(1e3 / fps_s, 3), 'stage_ms': st, 'bitwise_equal': bool(same), 'frames': n_frames,
            'workload': '%d synthetic frames: 1242x375 RGB (1.4 MB) + ~120 k x 4 fp32 sweep (2 MB) per frame from pinned host '
                        'memory, %d-%d detections (mean %.1f), 224x224 8-bit crops, Fusion A; per frame H2D -> prep_points '
                        '(image frustum + 3D boxes, one read-back) -> crop_resize_u8; per pair TrackingNet.forward -> '
                        'scores_for_solver (one packed D2H); no solver time' % (n_frames, ndet.min(), ndet.max(), ndet.mean())}


def main():
    args = parse_args()
    if args.rccl_world1_child:
        torch.cuda.set_device(args.device)
        print('RCCL_WORLD1 ' + json.dumps(rccl_world1_child(torch.device('cuda', args.device))), flush=True)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args)  # does not return
    claim_stdout()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    fusion, aff, sm, N, M, S, pts, gold_name = WORKLOADS[args.workload]
    # the global batch is G pairs (default world * --pairs); rank r owns the contiguous shard [lo, hi) of B pairs
    G = args.global_pairs if args.global_pairs is not None else world * args.pairs
    lo, hi = shard_range(G, rank, world)
    B = hi - lo
    even = G % world == 0   # equal shards: the flat gather needs no length exchange

    if args.dry:
        import torch.distributed as dist
        dev = torch.device('cpu')
        if world > 1:
            dist.init_process_group('gloo')

        def barrier():
            if world > 1:
                dist.barrier()

        def step():
            res = stub_results(hi - lo, N, M, 1000 + lo)
            return res if args.no_gather else gather_results(res, same_layout=even)

        place = bind_to_gpu_numa_node(None, enable=not args.no_bind)  # no GPU in a dry run: reports the affinity it found
        for _ in range(args.warmup):
            step()
        dt_own, res = time_steps(step, args.steps, barrier)
        dt = max_over_ranks(dt_own, world, dev)
        mine = dict(ms_per_step=round(dt_own / args.steps * 1e3, 3), pairs=hi - lo, first_pair=lo, gather_us_per_step=None,
                    sclk_mhz=None, sclk_vs_solo=None, power_w=None, **place)
        allr = [mine]
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        per_rank = {k: [x[k] for x in allr] for k in mine}
        # every rank checks the WHOLE gathered list: global (rank-major = pair-index) order, nothing lost or doubled
        ok = args.no_gather or (len(res) == G and all(
            torch.equal(res[i][1][0], stub_results(1, N, M, 1000 + i)[0][1][0]) and
            torch.equal(res[i][3], stub_results(1, N, M, 1000 + i)[0][3]) for i in range(G)))
        oks = [bool(ok)]
        if world > 1:
            oks = [None] * world
            dist.all_gather_object(oks, bool(ok))
        if rank == 0:
            out = dry_record(args, world, G, dt, per_rank, all(oks), place)
            emit(out, world, args.detail)
        if world > 1:
            dist.destroy_process_group()
        if not ok:
            raise SystemExit('dry run: gathered results are wrong')
        return

    from mmmot_amd import TrackingNet
    from mmmot_amd.weights import init_module
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    place = bind_to_gpu_numa_node(local_rank, enable=not args.no_bind)
    telemetry = Telemetry(local_rank)
    if args.latency_only:
        emit_line(json.dumps(latency_b1(dev, args.trunk)))
        return
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        if world == 1 and 'MASTER_ADDR' not in os.environ:
            init_single_rank_env()
        dist.init_process_group('nccl', device_id=dev)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def build_workload(name, Bw, rows, first, total=None):
        """model + plan + device-resident inputs of `Bw` frame pairs (global pair indices first .. first + Bw - 1)"""
        fusion_, aff_, sm_, N_, M_, S_, pts_, gold_ = WORKLOADS[name]
        mdl = TrackingNet(**dict(BASE_KW, score_fusion_arch=fusion_, affinity_op=aff_, softmax_mode=sm_))
        init_module(mdl, seed=0)
        mdl.eval().to(dev)
        seed0 = SEED0.get(name, 1000)
        # synthetic batch: distinct seeds per global pair index; inputs resident in HBM before timing.  `--input-sets`
        # batches of the same shape (one plan: the point counts are fixed) with disjoint seeds; set 0 is the batch the
        # reference goldens / the CPU baseline look at
        need_img, need_pts = (0 in rows) or (2 in rows), (1 in rows) or (2 in rows)
        sets, ins_ = [], None
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max(1, min(16, len(os.sched_getaffinity(0)))))  # one seeded generator per pair
        for k in range(max(args.input_sets, 1)):
            ik = list(pool.map(lambda i: make_pair(N_, M_, S_, pts_, seed=seed0 + 100003 * k + i), range(first, first + Bw)))
            sets.append((torch.cat([x[0] for x in ik]).to(dev) if need_img else None,
                         torch.cat([x[1]['points'].reshape(-1, 3) for x in ik]).to(dev) if need_pts else None))
            if k == 0:
                ins_ = ik
            else:
                assert all(torch.equal(a[1]['points_split'], b[1]['points_split']) for a, b in zip(ik, ins_))
        pool.shutdown()
        samples = [([N_, M_], x[1]['points_split'].reshape(-1).long().numpy() if need_pts else None) for x in ins_]
        plan_ = mdl.make_plan(samples, S_, rows=rows)
        torch.cuda.synchronize()
        set_bytes = sum(t.numel() * t.element_size() for t in sets[0] if t is not None)
        return dict(name=name, model=mdl, plan=plan_, sets=sets, ins=ins_, rows=rows, B=Bw, gold=gold_,
                    G=total if total is not None else Bw * world,
                    shape=(N_, M_, S_, pts_, fusion_), input_mb=round(len(sets) * set_bytes / 2 ** 20, 1))

    def run_leg(wl, trunk, steps, warmup, graph=False, profile=False):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize brackets, max over ranks."""
        mdl, plan_, sets = wl['model'], wl['plan'], wl['sets']
        mdl.set_trunk(trunk)
        eng = mdl.engine()
        # input rotation: call c of a phase of `n` calls runs set (n - 1 - c) % len(sets) - the LAST one runs set 0
        rot = {'n': 0, 'c': 0}

        def next_set():
            k = (rot['n'] - 1 - rot['c']) % len(sets) if rot['n'] else 0
            rot['c'] += 1
            return sets[k]

        def phase(n):
            rot['n'], rot['c'] = n, 0

        gather_ev = []  # HIP events around the result gather of every timed step (N-rank runs: where the time goes)

        def gathered(res):
            if args.no_gather:
                return res
            if not dist_on:
                return gather_results(res, same_layout=True)  # trivial for one process
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = gather_results(res, same_layout=(wl['G'] == wl['B'] * world), force=args.force_dist)
            e1.record()
            gather_ev.append((e0, e1))
            return res

        def step():
            crops_, points_ = next_set()
            return gathered(mdl.forward_batch(plan_, crops_, points_))

        phase(warmup)
        for _ in range(warmup):
            step()
        if graph:
            # the C-ABI entry points only launch (no allocation, no synchronisation): the whole step is capturable
            graphed = mdl.capture(plan_, sets[0][0], sets[0][1])

            def step():  # noqa: F811
                crops_, points_ = next_set()
                return gathered(graphed(crops_, points_))  # copies the set into the captured input buffers
            phase(1)
            step()
        eng.conv_events = []
        del gather_ev[:]
        phase(steps)
        with telemetry:
            dt, res = time_steps(step, steps, barrier)
        tele = telemetry.result()
        events, eng.conv_events = eng.conv_events, None
        dt_own = dt
        dt = max_over_ranks(dt, world, dev)
        roof, layers = roofline_of(trunk, events, eng, wl['B'], wl['name'], dt, wl['rows'])
        N_, M_, S_, pts_, fusion_ = wl['shape']
        value = steps * wl['G'] / dt
        fexec = executed_flops_per_pair(N_, M_, S_, (N_ + M_) * pts_, fusion_, wl['rows'],
                                        pn_gram=(trunk != 'f32' and getattr(eng, 'pn_gram', True)))
        leg = {'value': round(value, 4), 'ms_per_step': round(dt / steps * 1e3, 3), 'dtype': trunk, 'roofline': roof,
               'exec_gflop_per_pair': round(fexec / 1e9, 1),
               'exec_tflops_equiv_per_gpu': round(fexec * value / 1e12 / world, 2)}
        if trunk == 'f16x3':
            leg['whole_step_frac_of_f16x3_peak'] = round(fexec * value / 1e12 / world / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
        leg['inputs'] = '%d sets x %d pairs rotated over the steps, %.0f MiB resident' % (len(sets), wl['B'], wl['input_mb'])
        leg['telemetry'] = tele  # this rank's sustained shader clock / package power over the timed steps (hwmon)
        if profile and not graph:
            # per-launch-class device time of PROFILE_STEPS extra (untimed) steps: HIP events around every operator call.
            # EVERY rank runs them (a step of an N-rank run contains the result gather, a collective: a rank that took
            # more steps than the others would hang the job); rank 0's table is the one reported
            from mmmot_amd.profiler import LaunchProfiler
            phase(PROFILE_STEPS)
            with LaunchProfiler(eng.ops, f32=(trunk == 'f32')) as prof:
                for _ in range(PROFILE_STEPS):
                    step()
            if rank == 0:
                # compact rows (the JSON line stays one line): [class, launches/step, ms/step, bound, achieved, frac]
                summ = prof.summary(steps=PROFILE_STEPS, top=24)
                leg['kernels'] = {'device_ms_per_step': summ['device_ms_per_step'],
                                  'rows': [[r['class'], r['launches_per_step'], r['ms_per_step'], r['bound'],
                                            r.get('achieved'), r.get('frac')] for r in summ['classes']]}
            else:
                torch.cuda.synchronize()
        if dist_on:
            # every rank's own wall time per step and the device time of its result gather (pack + RCCL all_gather +
            # unpack, HIP events): `value` uses the MAX over ranks; this shows which rank and which part set it
            import torch.distributed as dist
            gus = (sum(e0.elapsed_time(e1) for e0, e1 in gather_ev) / max(len(gather_ev), 1)) * 1e3 if gather_ev else 0.0
            solo_clk = (solo or {}).get('sclk_mhz')
            mine = dict(ms_per_step=round(dt_own / steps * 1e3, 3), pairs=wl['B'], gather_us_per_step=round(gus, 1),
                        sclk_mhz=tele['sclk_mhz'], power_w=tele['power_w'],
                        # this rank's sustained clock over rank 0's clock running ALONE on the node (solo_leg): a
                        # power- or thermally-capped 8-GPU node shows up here as a ratio < 1 next to the scaling number
                        sclk_vs_solo=round(tele['sclk_mhz'] / solo_clk, 4) if tele['sclk_mhz'] and solo_clk else None,
                        **place)
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            leg['per_rank'] = {k: [x[k] for x in allr] for k in mine}
        if rank == 0 and wl['gold'] is not None:
            leg['linf_vs_reference_golden'] = golden_linf(res[0], wl['gold'], wl['rows'])
        return leg, res, layers

    def solo_leg(wl, trunk, n):
        """N-rank runs: rank 0 runs `n` ungathered steps ALONE while the other ranks wait at a barrier - its shader
        clock, package power and ms per step without neighbours drawing power on the same node (reference for
        per_rank.sclk_vs_solo and for the 1-GPU point of the scaling curve measured in the same process)."""
        import torch.distributed as dist
        mdl = wl['model']
        mdl.set_trunk(trunk)
        info = [None]
        barrier()
        if rank == 0:
            for k in range(n + 2):
                if k == 2:
                    torch.cuda.synchronize()
                    telemetry.__enter__()
                    t0 = time.perf_counter()
                mdl.forward_batch(wl['plan'], *wl['sets'][k % len(wl['sets'])])
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            telemetry.__exit__()
            t = telemetry.result()
            info[0] = {'ms_per_step': round(dt1 / n * 1e3, 3), 'pairs': wl['B'], 'steps': n, 'sclk_mhz': t['sclk_mhz'],
                       'power_w': t['power_w']}
        barrier()
        dist.broadcast_object_list(info, src=0)
        return info[0]

    rows = tuple(int(r) for r in args.rows.split(',') if r != '')
    head_wl = build_workload(args.workload, B, rows, lo, total=G)
    solo = solo_leg(head_wl, args.trunk, max(args.steps, 20)) if dist_on else None
    model, ins = head_wl['model'], head_wl['ins']
    do_prof = not args.no_profile
    kernels = {}
    head, res, layers = run_leg(head_wl, args.trunk, args.steps, args.warmup, graph=args.graph, profile=do_prof)
    if 'kernels' in head:
        kernels['headline_%s_%s' % (args.workload, args.trunk)] = head.pop('kernels')
    value = head['value']
    fref = reference_flops_per_pair(N, M, S, (N + M) * pts, fusion)
    out = {
        'metric': METRIC, 'value': value, 'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.trunk, 'data': 'synthetic',
        'config': {'workload': '%s: Fusion %s, %s/%s, N=M=%d (%d crops of %dx%d), %d pts/det; %d pairs/step/GPU%s' % (
            args.workload, fusion, aff, sm, N, N + M, S, S, pts, B,
            '' if rows == (0, 1, 2) else '; modality rows %s' % (rows,)), 'pairs_per_step_per_gpu': B, 'trunk': args.trunk,
            'parallelism': 'sample-sharded x%d, flat all_gather of scores' % world, 'hipgraph': bool(args.graph),
            'rccl': bool(dist_on)},
        'roofline': head['roofline'],
        'end_to_end': {
            'ref_gflop_per_pair': round(fref / 1e9, 1), 'ref_tflops_equiv': round(fref * value / 1e12 / world, 2),
            'exec_gflop_per_pair': head['exec_gflop_per_pair'], 'exec_tflops_equiv': head['exec_tflops_equiv_per_gpu'],
            'whole_step_frac_of_f16x3_peak': head.get('whole_step_frac_of_f16x3_peak'),
            'basis': 'F_ref = reference-as-written FLOPs (SURVEY 8d); F_exec = FLOPs of the math executed (STN trunks and '
                     'the 1088->512 broadcast eliminated, the 128 x 128 and 64 x 64 Gram matrices behind two GroupNorms added); '
                     'whole-step fraction = F_exec x pairs/s / GPU / (2500/3) - north_star target >= 0.50'},
        'parity': {'tolerance': 1e-3},
        'inputs': head['inputs'],
        'host': dict(place, telemetry=head['telemetry'],
                     note='rank 0: NUMA node / cores the rank was bound to (sysfs numa_node / local_cpulist of its GPU), '
                          'mean shader clock (MHz) and package power (W) over the timed steps (hwmon, 4 samples/s)'),
    }
    if 'per_rank' in head:
        out['per_rank'] = head['per_rank']
    if solo is not None:
        out['solo'] = solo
    if head.get('linf_vs_reference_golden') is not None:
        out['parity']['linf_vs_reference_golden'] = head['linf_vs_reference_golden']
        out['parity']['golden'] = ('tests/golden/%s.npz (output of the imported reference on the first pair of the batch)'
                                   % head_wl['gold'])
    if rank == 0:
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'bench_conv_layers_n%d.json' % world), 'w') as f:
                json.dump(layers, f, indent=1)
        except OSError:
            pass

    # ---- CPU baseline + parity on the same inputs (rank 0, single-GPU runs only) ----
    if rank == 0 and world == 1 and args.cpu_pairs > 0 and rows == (0, 1, 2):
        base, linf = cpu_baseline(model, ins, res, fusion, aff, sm, N, M, max(args.cpu_pairs, 1))
        out['cpu_baseline'] = base
        out['parity']['linf_vs_cpu_oracle'] = linf

    # ---- the other arithmetic legs, same run, same step counts ----
    extra = args.extra_trunks
    if extra is None:
        extra = {'f16x3': 'f16q8', 'f16q8': 'f16x3'}.get(args.trunk, 'none')
    out['extra'] = {}
    for t in [t for t in extra.split(',') if t and t != 'none' and t != args.trunk]:
        leg, _, _ = run_leg(head_wl, t, args.steps, args.warmup)
        out['extra'][t] = leg
    if not args.graph and extra != 'none':
        # the headline arithmetic again, the step captured once in a hipGraph and replayed (same K / W; no per-launch
        # HIP events inside a graph, so no roofline entry for this leg)
        leg, _, _ = run_leg(head_wl, args.trunk, args.steps, args.warmup, graph=True)
        out['extra'][args.trunk + '_hipgraph'] = {k: leg[k] for k in ('value', 'ms_per_step', 'dtype', 'linf_vs_reference_golden')
                                                  if k in leg}
    del head_wl, model, ins, res
    torch.cuda.empty_cache()

    # ---- the other BASELINE.json configs at their batch sizes (same arithmetic, same K / W) ----
    if not args.no_workloads and args.workload == 'cfg3' and rows == (0, 1, 2):
        wls = {}
        for key, name, Bw, wrows in EXTRA_WORKLOADS:
            wl = build_workload(name, Bw, wrows, rank * Bw)
            leg, _, _ = run_leg(wl, args.trunk, args.steps, args.warmup, profile=do_prof)
            if 'kernels' in leg:
                kernels[key] = leg.pop('kernels')
            N_, M_, S_, pts_, fusion_ = wl['shape']
            leg['config'] = '%s: Fusion %s, N=M=%d, %dx%d crops, %d pts/det, %d pairs/step/GPU, modality rows %s' % (
                name, fusion_, N_, S_, S_, pts_, Bw, wrows)
            leg['unit'] = 'frame-pairs/s'
            wls[key] = leg
            del wl
            torch.cuda.empty_cache()
        out['extra']['workloads'] = wls

    # ---- the collective path of the N-GPU run, executed on this one GPU (RCCL, world size 1) ----
    if rank == 0 and world == 1 and not dist_on and not args.no_workloads:
        out['extra']['rccl_world1'] = rccl_world1_leg(dev)

    if rank == 0 and world == 1 and not args.no_latency:
        out['extra']['latency'] = latency_b1(dev, args.trunk)

    if rank == 0 and world == 1 and not args.no_latency and not args.no_workloads:
        out['extra']['pipeline'] = pipeline_leg(dev, args.trunk)

    if rank == 0 and kernels:
        out['extra']['kernels'] = dict(
            kernels, columns=['launch class', 'launches per step', 'ms per step', 'bound', 'achieved', 'fraction of that peak'],
            how='HIP events (torch.cuda.Event on the launch stream) around every operator call of %d untimed steps behind each '
            'leg\'s timed region (mmmot_amd/profiler.py).  bound mfma: achieved = algorithmic TFLOP/s-equivalent against %.0f '
            '(f16 MFMA dense peak / 3); hbm: algorithmic TB/s against %.0f; latency: under 5 us of roofline time per launch.  '
            'device_ms_per_step = sum of the rows (compare the leg\'s ms_per_step)' % (PROFILE_STEPS, PEAK_F16_MFMA_TFLOPS / 3.0, 8.0))
    if rank == 0 and world == 1 and do_prof and not args.no_workloads:
        out['extra']['prep'] = prep_leg(dev)

    if rank == 0:
        emit(out, world, args.detail)
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
