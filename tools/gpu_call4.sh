#!/bin/bash
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05a/pytest_gpu_full.log 2>&1
tail -n 3 gpurun_out/r05a/pytest_gpu_full.log
bash tools/measure_round.sh r05a
