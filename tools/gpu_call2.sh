#!/bin/bash
# round-5 batch 2: kernel tests of the changed small kernels, whole-forward A/B against the round-4 library, the new
# full-size goldens, the bench with its new fields
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_points_gpu.py tests/test_robust_gpu.py -x -q -m gpu > $O/pytest_kernels.log 2>&1
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_size and (calibrated or seed1)" -s > $O/pytest_new_goldens.log 2>&1
timeout 1200 python tools/ab_forward.py --legs cfg2:32 cfg4:16 cfg3:8 cfg3:16:1 --rounds 2 --steps 12 > $O/ab_forward.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err
tail -n 4 $O/pytest_kernels.log; grep -E "full-size|passed|failed" $O/pytest_new_goldens.log | tail -12; cat $O/ab_forward.log; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c2/bench_default.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['end_to_end']['whole_step_frac_of_f16x3_peak'], d['inputs'], d['host'])
for k, v in d['extra']['workloads'].items():
    print(k, v['value'], v['ms_per_step'], v.get('whole_step_frac_of_f16x3_peak'), v['telemetry'])
print(json.dumps(d['extra'].get('prep'), indent=0))
for leg, s in d['extra']['kernels'].items():
    if leg == 'how': continue
    print(leg, 'device ms/step', s['device_ms_per_step'])
    for r in s['classes'][:12]:
        print('   ', r)
PY
