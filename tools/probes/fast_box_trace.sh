#!/bin/bash
# On a box of the "fast" kind (the unstaggered library faster than the staggered one) record the queue trace of
# a step with the lock-step graph, to compare with the slow kind's (profiles/r05_queue_trace_b1_graph.txt).
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/fastbox; mkdir -p $OUT
M4D_STAGGER_AUTOTUNE=0 tools/ab_bench.sh 2 "M4D_WINO6_STAGGER_US=0" "M4D_WINO6_STAGGER_US=9" > $OUT/ab_$$.txt 2>&1
cat $OUT/ab_$$.txt | tail -2
A=$(grep "STAGGER_US=0" $OUT/ab_$$.txt | awk '{print $3}'); B=$(grep "STAGGER_US=9" $OUT/ab_$$.txt | awk '{print $3}')
KIND=$(python -c "print('fast' if float('$A') > float('$B') else 'slow')")
echo "kind: $KIND"
for US in 0 9; do
  rm -rf /tmp/pq_$US
  M4D_STAGGER_AUTOTUNE=0 M4D_WINO6_STAGGER_US=$US timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq_$US -o t -- python bench.py --steps 20 --no-cpu-baseline --no-kernel-timing --no-configs2 > /dev/null 2>&1
  T=$(find /tmp/pq_$US -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/queue_trace.py "$T" > $OUT/queue_trace_${KIND}_us${US}_$$.txt 2> /dev/null
done
ls -la $OUT | tail -5
