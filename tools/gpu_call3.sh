#!/bin/bash
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_vgg_gpu.py tests/test_gemm_f16_gpu.py -x -q -m gpu > $O/pytest_kernels.log 2>&1
timeout 600 python tools/bench_rows_gemm.py --rows 1572864 --k 512 --n 128 --variants 1 2 --sustain 40 > $O/rows_gemm_n128.log 2>&1
timeout 600 python tools/bench_rows_gemm.py --rows 196608 --k 512 --n 128 --variants 1 2 --sustain 100 >> $O/rows_gemm_n128.log 2>&1
timeout 600 python tools/bench_rows_gemm.py --rows 32768 --k 512 --n 128 --variants 1 2 --sustain 100 >> $O/rows_gemm_n128.log 2>&1
timeout 1200 python tools/ab_forward.py --legs cfg2:32 cfg3:32:1 cfg4:8 --rounds 3 --steps 12 > $O/ab_forward.log 2>&1
tail -n 3 $O/pytest_kernels.log; cat $O/rows_gemm_n128.log | grep -v amdgpu.ids; cat $O/ab_forward.log
