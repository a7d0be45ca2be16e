#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, counters only) of the level-1 DSCV /
# SNCV kernels at the given batch; writes profiles-ready text.  usage: tools/pmc_traffic.sh <batch>
set -u
B=${1:-1}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
OUT=gpurun_out/pmc/traffic_l1_b${B}.txt
: > $OUT
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$CNT
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/pmct_$CNT -o p -- python tools/bench_kernels.py --batch $B --iters 5 --which dscv,sncv > /tmp/pmct_$CNT.log 2>&1
  f=$(find /tmp/pmct_$CNT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/summarize_pmc.py "$f" | grep -E "dscv_wave_kernel<4, 4, 9>|sncv7_kernel<16, 1, 32, 8>" >> $OUT
done
cat $OUT
