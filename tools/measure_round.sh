#!/bin/bash
# One gpurun call that refreshes the measurements kept under profiles/ (run from the repo root on the GPU box):
#   tools/measure_round.sh <tag>      -> gpurun_out/<tag>/...
# rocprofv3 --pmc passes are separate runs without trace domains (the pool refuses mixed runs).
tag=${1:-final}
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the driver's command line (defaults: cfg3, 16 pairs/step, f16x3; extra: f16q8, hipGraph, the other BASELINE configs, RCCL, latency)
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
Q="--cpu-pairs 0 --extra-trunks none --no-latency --no-workloads"
rm -rf /tmp/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py --steps 4 --warmup 1 $Q > $O/rocprof_stats_run.log 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/prof_stats -name "*_results.db" | head -1) > $O/rocprofv3_kernel_stats_cfg3_pairs16_f16x3.txt 2>&1
for trunk in f16x3 f16q8; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    if [ $trunk = f16q8 ] && [ $n = SQ_VALU_MFMA_BUSY_CYCLES ]; then continue; fi
    rm -rf /tmp/prof_pmc
    timeout 300 rocprofv3 --pmc $c -d /tmp/prof_pmc -- python $R/bench.py --steps 2 --warmup 1 --pairs 1 --trunk $trunk $Q > $O/rocprof_pmc_${n}_$trunk.log 2>&1
    python $R/tools/rocpd_summary.py pmc $(find /tmp/prof_pmc -name "*_results.db" | head -1) > $O/rocprofv3_pmc_${n}_cfg3_pairs1_$trunk.txt 2>&1
  done
done
rm -rf /tmp/prof_lat
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat -- python $R/bench.py --latency-only > $O/rocprof_latency_run.log 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/prof_lat -name "*_results.db" | head -1) > $O/rocprofv3_kernel_stats_latency_b1.txt 2>&1
timeout 300 python $R/tools/bench_backward.py > $O/bench_backward.log 2>&1
timeout 300 python $R/tools/bench_train.py > $O/bench_train.log 2>&1
timeout 300 python $R/tools/bench_conv_variants.py --rounds 4 --variants 11,15,18 > $O/conv_variants_winograd_proxy.log 2>&1
timeout 300 python $R/tools/bench_fused1.py > $O/bench_fused1.log 2>&1
timeout 300 python $R/tools/fused1_phase_timers.py > $O/fused1_phase_timers.log 2>&1
timeout 300 python $R/bench.py --steps 10 --warmup 3 --pairs 32 $Q > $O/bench_pairs32_probe.log 2>&1
cd $R
# the tree compiles from clean on the box (no prebuilt objects reused), then the smoke check runs on that build
( MMMOT_FORCE_BUILD=1 timeout 900 python -c "import time, __graft_entry__ as g; t = time.time(); print(g.build()); print('forced rebuild of every HIP source: %.0f s' % (time.time() - t)); g.smoke()" ) > $O/smoke_forced_build.log 2>&1
tail -c 1500 $O/bench_default.log; head -14 $O/rocprofv3_kernel_stats_cfg3_pairs16_f16x3.txt; tail -4 $O/smoke_forced_build.log
