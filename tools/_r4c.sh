O=gpurun_out/r4c; mkdir -p $O; cd /root/repo
Q="--cpu-pairs 0 --extra-trunks none --no-latency --no-workloads"
timeout 900 python -m pytest tests/test_gemm_f16_gpu.py tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.log 2>&1; tail -4 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_batch_sizes_gpu.py tests/test_parity_gpu.py -x -q > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log
for w in "cfg3 16" "cfg4 32" "cfg2 32"; do set -- $w; timeout 200 python bench.py --steps 8 --warmup 3 --workload $1 --pairs $2 $Q > $O/bench_$1.log 2>&1; python - <<PY
import json
l=[x for x in open('$O/bench_$1.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['whole_step_frac_of_f16x3_peak'], d['parity']['linf_vs_reference_golden'])
else:
    print(open('$O/bench_$1.log').read()[-1500:])
PY
done
cd /tmp && export TMPDIR=/tmp
for w in "cfg4 32" "cfg3 16"; do set -- $w; rm -rf /tmp/prof_$1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -- python /root/repo/bench.py --steps 3 --warmup 1 --workload $1 --pairs $2 $Q > /root/repo/$O/rocprof_$1.log 2>&1; python /root/repo/tools/rocpd_summary.py stats $(find /tmp/prof_$1 -name "*_results.db" | head -1) > /root/repo/$O/kernel_stats_$1.txt 2>&1; grep -E "gemm_wide|gemm_rows|gram_rows|segment_mean_kernel|total" /root/repo/$O/kernel_stats_$1.txt | cut -c1-150; done
