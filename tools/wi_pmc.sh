#!/bin/bash
# PMC passes over the K = 64 column-sum pass of mmmot_gemm_ares (tools/bench_ares.py): SQ issue / wait split, MFMA busy.
#   tools/wi_pmc.sh <out dir> [variant, default 2]   (on the GPU box; one counter group per run, no trace domains)
out=${1:-gpurun_out/wi_pmc}; var=${2:-2}; R=$PWD
mkdir -p $out
export TMPDIR=/tmp
cmd="python $R/tools/bench_ares.py --rows 8388608 --k 64 --n 512 --modes colsum --variants $var"
pass() {  # name, counters
  local name=$1 ctr=$2
  rm -rf /tmp/wip_$name
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d /tmp/wip_$name -- $cmd > $R/$out/run_$name.log 2>&1 )
  python $R/tools/rocpd_summary.py pmc $(find /tmp/wip_$name -name "*_results.db" | head -1) 2>&1 | grep -i "wres\|^kernel" > $R/$out/pmc_$name.txt
}
pass sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
pass sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
pass grbm "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
pass tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
cat $out/pmc_*.txt
