#!/bin/bash
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1
tail -n 4 $O/pytest_gpu_full.log
timeout 900 python tools/ab_forward.py --legs cfg4:16 cfg2:32 cfg3:4 --rounds 2 --steps 12 > $O/ab_forward.log 2>&1
cat $O/ab_forward.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c5/bench_default.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['end_to_end']['whole_step_frac_of_f16x3_peak'], d['host']['telemetry'])
for k, v in d['extra']['workloads'].items():
    print(k, v['value'], v['ms_per_step'], v.get('whole_step_frac_of_f16x3_peak'))
for r in d['extra']['kernels']['cfg4_b32_per_gpu']['classes']:
    if r['class'].startswith(('GroupNorm', 'row dot', 'SkipPool')): print(r)
PY
