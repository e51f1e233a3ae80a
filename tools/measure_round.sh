#!/bin/bash
# One gpurun call that refreshes the measurements kept under profiles/ (run from the repo root on the GPU box):
#   tools/measure_round.sh <tag>      -> gpurun_out/<tag>/...
# rocprofv3 --pmc passes are separate runs without trace domains (the pool refuses mixed runs).
tag=${1:-final}
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (--no-profile: the rocprof passes count exactly warm-up + timed steps; extra.kernels comes from the default run above)
Q="--cpu-pairs 0 --extra-trunks none --no-latency --no-workloads --no-profile"
# the driver's command line (defaults: cfg3, 16 pairs/step, f16x3; extra: f16q8, hipGraph, the other BASELINE configs, RCCL, latency)
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; cp $R/gpurun_out/bench_detail_n1.json $O/bench_detail_n1.json
# kernel tables: the headline workload and the other BASELINE configs at their batch sizes
stats() {  # name, bench arguments
  local name=$1; shift
  rm -rf /tmp/prof_stats
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py --steps 4 --warmup 1 $Q "$@" > $O/rocprof_stats_$name.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find /tmp/prof_stats -name "*_results.db" | head -1) > $O/rocprofv3_kernel_stats_$name.txt 2>&1
}
stats cfg3_pairs16_f16x3
stats cfg2_b32_f16x3 --workload cfg2 --pairs 32
stats cfg4_b32_f16x3 --workload cfg4 --pairs 32
stats cfg5_lidar_b32_f16x3 --rows 1 --pairs 32
stats cfg5_image_b8_f16x3 --rows 0 --pairs 8
# counter passes (3 steps each): HBM traffic of every config, matrix-core busy cycles of the headline one
pmc() {  # name, counters, bench arguments
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_pmc
  timeout 300 rocprofv3 --pmc $ctr -d /tmp/prof_pmc -- python $R/bench.py --steps 2 --warmup 1 $Q "$@" > $O/rocprof_pmc_$name.log 2>&1
  python $R/tools/rocpd_summary.py pmc $(find /tmp/prof_pmc -name "*_results.db" | head -1) > $O/rocprofv3_pmc_$name.txt 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ${c}_cfg3_pairs1_f16x3 $c --pairs 1
  pmc ${c}_cfg3_pairs1_f16q8 $c --pairs 1 --trunk f16q8
  pmc ${c}_cfg2_pairs4_f16x3 $c --workload cfg2 --pairs 4
  pmc ${c}_cfg4_pairs2_f16x3 $c --workload cfg4 --pairs 2
  pmc ${c}_cfg5_lidar_pairs2_f16x3 $c --rows 1 --pairs 2
done
pmc SQ_VALU_MFMA_BUSY_CYCLES_cfg3_pairs1_f16x3 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --pairs 1
pmc SQ_VALU_MFMA_BUSY_CYCLES_cfg4_pairs2_f16x3 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload cfg4 --pairs 2
rm -rf /tmp/prof_lat
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat -- python $R/bench.py --latency-only > $O/rocprof_latency_run.log 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/prof_lat -name "*_results.db" | head -1) > $O/rocprofv3_kernel_stats_latency_b1.txt 2>&1
# kernel table of the whole-network training step (cfg1 shape) at HEAD
rm -rf /tmp/prof_train
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $R/tools/profile_train_step.py --whole > $O/rocprof_train_run.log 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/prof_train -name "*_results.db" | head -1) > $O/rocprofv3_kernel_stats_train.txt 2>&1
python $R/tools/rocpd_summary.py detail $(find /tmp/prof_train -name "*_results.db" | head -1) 60 > $O/rocprofv3_kernel_detail_train.txt 2>&1
timeout 300 python $R/tools/train_segment_calls.py 2>&1 | grep "calls  nseg" > $O/train_segment_calls.txt
timeout 300 python $R/tools/bench_backward.py > $O/bench_backward.log 2>&1
timeout 300 python $R/tools/bench_train.py > $O/bench_train.log 2>&1
timeout 300 python $R/tools/bench_conv_small_maps.py > $O/conv_small_maps_ab.log 2>&1
timeout 300 python $R/tools/bench_ares.py --rows 4194304 --n 1024 --variants 1 2 > $O/bench_ares_variants.log 2>&1
cd $R
# the tree compiles from clean on the box (no prebuilt objects reused), then the smoke check runs on that build
( MMMOT_FORCE_BUILD=1 timeout 900 python -c "import time, __graft_entry__ as g; t = time.time(); print(g.build()); print('forced rebuild of every HIP source: %.0f s' % (time.time() - t)); g.smoke()" ) > $O/smoke_forced_build.log 2>&1
cat $O/bench_default.out | head -c 1200; head -14 $O/rocprofv3_kernel_stats_cfg3_pairs16_f16x3.txt; tail -4 $O/smoke_forced_build.log
