#!/bin/bash
# PMC passes over the wide row GEMM (tools/bench_rows_gemm.py, variant 2): SQ issue/wait split, L1 -> L2 requests, LDS conflicts.
#   tools/gw_pmc.sh <out dir> [kernel substring, default gemm_wide]   (on the GPU box; one counter group per run, no trace domains)
out=${1:-gpurun_out/gw_pmc}; pat=${2:-gemm_wide}; R=$PWD
mkdir -p $out
export TMPDIR=/tmp
pass() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/gwp_$name
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d /tmp/gwp_$name -- "$@" > $R/$out/run_$name.log 2>&1 )
  python $R/tools/rocpd_summary.py pmc $(find /tmp/gwp_$name -name "*_results.db" | head -1) 2>&1 | grep -i "$pat\|^kernel" > $R/$out/pmc_$name.txt
}
for mode in pair norm; do
  if [ $mode = pair ]; then cmd="python $R/tools/bench_rows_gemm.py --pair 128 --rows 393216 --n 1024 --variants ${VAR:-4} --sustain 2"
  else cmd="python $R/tools/bench_rows_gemm.py --rows 393216 --groups 24 --n 512 --variants ${VAR:-4} --sustain 2"; fi
  pass ${mode}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" $cmd
  pass ${mode}_sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $cmd
  pass ${mode}_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" $cmd
  pass ${mode}_tcp2 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" $cmd
  pass ${mode}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_TAG_STALL_sum" $cmd
  pass ${mode}_grbm "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" $cmd
done
cat $out/pmc_*.txt
