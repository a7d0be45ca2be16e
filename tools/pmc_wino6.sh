#!/bin/bash
# rocprofv3 PMC passes (counters only) over the bf16-split Winograd kernel on the level-1 128->128 layer.
# usage: bash tools/pmc_wino6.sh [variant]   (variant 2 = the wide kernel m4d_wino6w.hip -> gpurun_out/pmc/wino6_v2.txt)
set -u
V=${1:-0}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
OUT=gpurun_out/pmc/wino6.txt; [ $V != 0 ] && OUT=gpurun_out/pmc/wino6_v$V.txt; : > $OUT
i=0
for CNT in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcw_$i
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/pmcw_$i -o p -- python tools/bench_conv_one.py --iters 3 --winograd 6 --variant $V > /tmp/pmcw_$i.log 2>&1
  f=$(find /tmp/pmcw_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" >> $OUT; else echo "pass $i ($CNT) produced no csv" >> $OUT; tail -3 /tmp/pmcw_$i.log >> $OUT; fi
done
cat $OUT
