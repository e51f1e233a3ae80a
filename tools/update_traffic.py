#!/usr/bin/env python
"""profiles/traffic.json from the committed PMC summaries of a round: HBM bytes of the dominant kernel per frame pair, for
the headline workload and the other BASELINE configs (bench.py reads it for `roofline.traffic`).

    python tools/update_traffic.py profiles/r04

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch (tools/rocpd_summary.py prints mean and sum over the run's dispatches);
the passes run `bench.py --steps 2 --warmup 1 --pairs P` = 3 steps of P pairs, so sum / (3 P) = one pair.  gfx950's FETCH_SIZE
counts half of the bytes of 16 B/lane reads (calibrated with a copy kernel: profiles/README.md), hence the factor 2."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3
TRUNK = ('conv3x3_hl16_patch_kernel',)
LIDAR = ('gemm_wreg128_kernel', 'gemm_ares_kernel')
# key in traffic.json, file tag of the passes, pairs per step of the passes, kernel name prefixes, launches per step, label, command
CASES = [
    ('cfg3/f16x3', 'cfg3_pairs1_f16x3', 1, TRUNK, 12, 'conv3x3_hl16_patch_kernel (12 launches per pair-batch; the first one also computes conv1_1)',
     '--pairs 1'),
    ('cfg3/f16q8', 'cfg3_pairs1_f16q8', 1, TRUNK, 12, 'conv3x3_hl16_patch_kernel<..., Q8> (12 launches per pair-batch; the first one also computes conv1_1)',
     '--pairs 1 --trunk f16q8'),
    ('cfg2/f16x3', 'cfg2_pairs4_f16x3', 4, TRUNK, 12, 'conv3x3_hl16_patch_kernel at 64-pixel crops (12 launches; conv5_x in the whole-map geometry)',
     '--workload cfg2 --pairs 4'),
    ('cfg4/f16x3', 'cfg4_pairs2_f16x3', 2, TRUNK, 12, 'conv3x3_hl16_patch_kernel at 64-pixel crops, N = M = 128 (12 launches; conv5_x in the whole-map geometry)',
     '--workload cfg4 --pairs 2'),
    ('cfg3_lidar/f16x3', 'cfg5_lidar_pairs2_f16x3', 2, LIDAR, 1, 'PointNet conv5 128 -> 1024 consumer pass (gemm_wreg128_kernel; gemm_ares_kernel on small launches)',
     '--rows 1 --pairs 2'),
]


def kernel_kb(path, prefixes):
    """KB summed over every launch of the named kernels in the run (rows: kernel, counter, calls, mean / dispatch, sum, avg us)"""
    tot = 0.0
    for line in open(path):
        if not line.startswith(prefixes):
            continue
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            if ' ' + ctr + ' ' in line:
                tot += float(line.split(ctr)[1].split()[2])
    return tot


def main():
    d = sys.argv[1].rstrip('/')
    out = {}
    for key, tag, pairs, prefixes, launches, label, cmd in CASES:
        f = os.path.join(d, 'rocprofv3_pmc_FETCH_SIZE_%s.txt' % tag)
        w = os.path.join(d, 'rocprofv3_pmc_WRITE_SIZE_%s.txt' % tag)
        if not (os.path.exists(os.path.join(ROOT, f)) and os.path.exists(os.path.join(ROOT, w))):
            print('missing passes for', key)
            continue
        fetch = kernel_kb(os.path.join(ROOT, f), prefixes) * 1024.0 * 2.0 / (STEPS * pairs)
        write = kernel_kb(os.path.join(ROOT, w), prefixes) * 1024.0 / (STEPS * pairs)
        out[key] = {
            'kernel': label,
            'fetch_bytes_per_pair': int(round(fetch)),
            'write_bytes_per_pair': int(round(write)),
            'launches_per_step': launches,
            # (short: bench.py repeats it in every leg of its one JSON line; the method is in profiles/README.md: FETCH_SIZE x 2
            # - gfx950 reports half of 16 B/lane reads - + WRITE_SIZE, KB per dispatch summed over the kernel's launches,
            # separate --pmc passes of `python bench.py --steps 2 --warmup 1 <cmd> ... --no-profile`, linear in pairs/step)
            'source': '%s x2 + %s (PMC passes at %d pair(s)/step: bench.py %s)' % (f, w, pairs, cmd),
        }
    with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
        fh.write('\n')
    for k, v in out.items():
        print(k, v['fetch_bytes_per_pair'], v['write_bytes_per_pair'])


if __name__ == '__main__':
    main()
