#!/bin/bash
# round-end GPU pass of round 5 (one gpurun call): full -m gpu suite, tools/measure_round.sh, the --force-dist step, old-vs-new library A/B
# (the A/B leg needs a libmmmot_hip.so built from the tree to compare with - e.g. a `git worktree` of the previous round - at $OLD_LIB)
mkdir -p gpurun_out/r05b
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r05b/pytest_gpu_full.log 2>&1
tail -n 3 gpurun_out/r05b/pytest_gpu_full.log
bash tools/measure_round.sh r05b > gpurun_out/r05b/measure_round.out 2>&1
tail -n 12 gpurun_out/r05b/measure_round.out
timeout 600 python bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --cpu-pairs 0 --extra-trunks none --no-latency --no-workloads > gpurun_out/r05b/bench_force_dist.log 2>&1
tail -c 1200 gpurun_out/r05b/bench_force_dist.log
timeout 900 python tools/ab_forward.py --lib-a "${OLD_LIB:-tools/_ab/libmmmot_hip_old.so}" --legs cfg4:16 cfg2:32 cfg3:8 cfg3:32:1 --rounds 2 --steps 12 > gpurun_out/r05b/ab_forward_vs_r04_library.log 2>&1
cat gpurun_out/r05b/ab_forward_vs_r04_library.log
