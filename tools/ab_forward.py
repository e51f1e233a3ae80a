#!/usr/bin/env python
"""Same-box A/B of two builds of libmmmot_hip.so on whole forwards: results bitwise, step times side by side.

    python tools/ab_forward.py --lib-a tools/_ab/libmmmot_hip_old.so [--lib-b mmmot_amd/libmmmot_hip.so] [--legs cfg2:8 ...]

Each leg "<workload>:<pairs>[:<rows>]" (workloads of bench.py; rows like "1" = LiDAR-only) runs in a child process per
library (MMMOT_LIB_PATH selects the build before mmmot_amd is imported), alternating A / B / A / B so that clock drift of
the box shows up as spread instead of as a difference.  The child prints ms per step and dumps every output tensor of the
batch; the parent compares the dumps of A and B bit for bit.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from mmmot_amd import TrackingNet
    from mmmot_amd.synth import make_pair
    from mmmot_amd.weights import init_module
    name, B, rows = a.workload, a.pairs, tuple(int(r) for r in a.rows.split(','))
    fusion, aff, sm, N, M, S, pts, _ = bench.WORKLOADS[name]
    dev = torch.device('cuda:0')
    mdl = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch=fusion, affinity_op=aff, softmax_mode=sm))
    init_module(mdl, seed=0)
    mdl.eval().to(dev)
    ins = [make_pair(N, M, S, pts, seed=1000 + i) for i in range(B)]
    need_img, need_pts = (0 in rows) or (2 in rows), (1 in rows) or (2 in rows)
    samples = [([N, M], x[1]['points_split'].reshape(-1).long().numpy() if need_pts else None) for x in ins]
    plan = mdl.make_plan(samples, S, rows=rows)
    crops = torch.cat([x[0] for x in ins]).to(dev) if need_img else None
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).to(dev) if need_pts else None
    mdl.set_trunk(a.trunk)
    for _ in range(a.warmup):
        res = mdl.forward_batch(plan, crops, points)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        res = mdl.forward_batch(plan, crops, points)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    flat = []
    for det, links, new, end in res:
        flat += [det.cpu(), new.cpu(), end.cpu()] + [l.cpu() for l in links]
    torch.save(flat, a.dump)
    print(json.dumps(dict(ms_per_step=round(ms, 4), pairs_per_s=round(B / ms * 1e3, 2))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib-a', default=os.path.join(ROOT, 'tools', '_ab', 'libmmmot_hip_old.so'))
    ap.add_argument('--lib-b', default=os.path.join(ROOT, 'mmmot_amd', 'libmmmot_hip.so'))
    ap.add_argument('--legs', nargs='*', default=['cfg2:8', 'cfg3:2', 'cfg4:2', 'cfg3:4:1'])
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--trunk', default='f16x3')
    # child mode
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--workload')
    ap.add_argument('--pairs', type=int)
    ap.add_argument('--rows', default='0,1,2')
    ap.add_argument('--dump')
    a = ap.parse_args()
    if a.child:
        return child(a)
    import torch
    tmp = tempfile.mkdtemp()
    for leg in a.legs:
        f = leg.split(':')
        wl, B, rows = f[0], int(f[1]), (f[2] if len(f) > 2 else '0,1,2')
        times = {'A': [], 'B': []}
        dumps = {}
        for r in range(a.rounds):
            for tag, lib in (('A', a.lib_a), ('B', a.lib_b)):
                dump = os.path.join(tmp, '%s_%s.pt' % (leg.replace(':', '_'), tag))
                env = dict(os.environ, MMMOT_LIB_PATH=os.path.abspath(lib), MMMOT_LIB_ALLOW_MISSING='1')
                out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--workload', wl, '--pairs', str(B),
                                      '--rows', rows, '--dump', dump, '--steps', str(a.steps), '--warmup', str(a.warmup),
                                      '--trunk', a.trunk], env=env, capture_output=True, text=True, timeout=900)
                if out.returncode != 0:
                    print('leg %s lib %s FAILED:\n%s' % (leg, tag, out.stderr[-2000:]), flush=True)
                    break
                times[tag].append(json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step'])
                dumps[tag] = dump
        same = None
        if 'A' in dumps and 'B' in dumps:
            ta, tb = torch.load(dumps['A']), torch.load(dumps['B'])
            same = len(ta) == len(tb) and all(torch.equal(x, y) for x, y in zip(ta, tb))
            worst = max((x - y).abs().max().item() for x, y in zip(ta, tb)) if not same and len(ta) == len(tb) else 0.0
        print('leg %-12s A %s ms   B %s ms   bitwise equal: %s%s' % (
            leg, times['A'], times['B'], same, '' if same or same is None else ' (max |A - B| = %.3e)' % worst), flush=True)


if __name__ == '__main__':
    main()
