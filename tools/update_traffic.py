#!/usr/bin/env python
"""profiles/traffic.json from the committed PMC summaries of a round: HBM bytes of the trunk kernel per frame pair.

    python tools/update_traffic.py profiles/r03

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch (tools/rocpd_summary.py prints mean and sum over the run's dispatches);
the passes run `bench.py --steps 2 --warmup 1 --pairs 1` = 3 steps of one pair, so sum / 3 = one pair.  gfx950's FETCH_SIZE
counts half of the bytes of 16 B/lane reads (calibrated with a copy kernel: profiles/README.md), hence the factor 2."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3


def trunk_kb(path):
    """KB summed over every trunk launch of the run (in f16q8 mode the first layers run the hl16 arithmetic)."""
    tot = 0.0
    for line in open(path):
        if not line.startswith('conv3x3_hl16_patch_kernel'):
            continue
        cols = line[line.index('>') + 1:].split()
        if cols[0] in ('FETCH_SIZE', 'WRITE_SIZE'):
            tot += float(cols[3])
    return tot


def main():
    d = sys.argv[1].rstrip('/')
    out = {}
    for mode, label in (('f16x3', 'conv3x3_hl16_patch_kernel (12 launches per pair-batch; the first one also computes conv1_1)'),
                        ('f16q8', 'conv3x3_hl16_patch_kernel<..., Q8> (12 launches per pair-batch; the first one also computes conv1_1)')):
        f = os.path.join(d, 'rocprofv3_pmc_FETCH_SIZE_cfg3_pairs1_%s.txt' % mode)
        w = os.path.join(d, 'rocprofv3_pmc_WRITE_SIZE_cfg3_pairs1_%s.txt' % mode)
        fetch = trunk_kb(os.path.join(ROOT, f)) * 1024.0 * 2.0 / STEPS
        write = trunk_kb(os.path.join(ROOT, w)) * 1024.0 / STEPS
        out['cfg3/' + mode] = {
            'kernel': label,
            'fetch_bytes_per_pair': int(round(fetch)),
            'write_bytes_per_pair': int(round(write)),
            'launches_per_step': 12,
            'source': ('%s (KB per dispatch summed over the 12 launches of one pair, x2: gfx950 FETCH_SIZE reports half of 16 B/lane '
                       'reads, calibrated in profiles/README.md) + %s; separate --pmc passes of `python bench.py --steps 2 --warmup 1 '
                       '--pairs 1 --trunk %s --cpu-pairs 0 --extra-trunks none --no-latency --no-workloads`; traffic scales linearly '
                       'with pairs/step' % (f, w, mode)),
        }
    with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
        fh.write('\n')
    for k, v in out.items():
        print(k, v['fetch_bytes_per_pair'], v['write_bytes_per_pair'])


if __name__ == '__main__':
    main()
