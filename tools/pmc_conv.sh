#!/bin/bash
# HBM traffic + MFMA busy counters of the level-1 128->128 convolution.  usage: tools/pmc_conv.sh <batch> [winograd 0|1|2]
set -u
B=${1:-1}
WG=${2:-0}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
OUT=gpurun_out/pmc/conv_l1_b${B}_wino${WG}.txt
: > $OUT
for CNT in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmcc
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/pmcc -o p -- python tools/bench_conv_one.py --batch $B --iters 5 --winograd $WG > /tmp/pmcc.log 2>&1
  f=$(find /tmp/pmcc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" | grep conv3x3 >> $OUT; else echo "no csv for $CNT" >> $OUT; tail -2 /tmp/pmcc.log >> $OUT; fi
done
cat $OUT
