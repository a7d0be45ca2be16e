#!/bin/bash
# rocprofv3 PMC passes (counters only: --pmc is never combined with tracing options here) for
# tools/bench_kernels.py.  usage: tools/pmc_pass.sh <out_prefix> <bench_kernels args...>
set -u
OUT=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
i=0
for CNT in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/pmc_$i -o p -- python tools/bench_kernels.py "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" >> gpurun_out/pmc/${OUT}.txt; else echo "pass $i ($CNT) produced no csv" >> gpurun_out/pmc/${OUT}.txt; tail -3 /tmp/pmc_$i.log >> gpurun_out/pmc/${OUT}.txt; fi
done
cat gpurun_out/pmc/${OUT}.txt
