#!/bin/bash
# round-5 experiment batch 1 (run from the repo root on the GPU box)
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
{
echo "== gemm_wide: weight layout / row pitch (NORM_RELU prologue, cfg4-sized) =="
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 512 --variants 2 --wsm --sustain 40
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 512 --variants 2 --wsm --sustain 40 --ldx 1024
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 512 --variants 2 --wsm --sustain 40 --ldx 1056
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 512 --variants 2 --wsm --sustain 40 --ldx 544
echo "== gemm_wide: PAIR prologue, M = 128 (cfg4) and 64 (cfg3) =="
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 1024 --variants 2 --wsm --sustain 30 --pair 128
timeout 300 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 1024 --variants 2 --wsm --sustain 30 --pair 128 --ldf 544
timeout 300 python tools/bench_rows_gemm.py --rows 196608 --k 512 --n 1024 --variants 2 --wsm --sustain 100 --pair 64
} > $O/rows_gemm.log 2>&1
{
echo "== A/B forward: old (round-4 kernels) vs new library =="
timeout 900 python tools/ab_forward.py --legs cfg2:32 cfg3:8:1 cfg3:2 --rounds 2 --steps 10
} > $O/ab_forward.log 2>&1
{
echo "== LiDAR-only batch size =="
for p in 8 16 32; do
  timeout 300 python bench.py --rows 1 --pairs $p --steps 20 --warmup 3 --cpu-pairs 0 --extra-trunks none --no-latency --no-workloads | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/step', d['config']['pairs_per_step_per_gpu'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['end_to_end']['whole_step_frac_of_f16x3_peak'])"
done
} > $O/lidar_batch.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gram or finalize" > $O/pytest_gram.log 2>&1
tail -n 30 $O/rows_gemm.log; cat $O/ab_forward.log; cat $O/lidar_batch.log; tail -n 3 $O/pytest_gram.log
