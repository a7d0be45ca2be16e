#!/bin/bash
# Round-end evidence: bench lines (batch 1 / 8 / 32), rocprofv3 kernel stats + step timeline of the batch-1 bench,
# PMC traffic.  usage (GPU box, repo root): bash tools/collect_round_profiles.sh r02   -> gpurun_out/<tag>/
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python tools/pmc_traffic.py --batch 1 > $OUT/pmc.log 2>&1
timeout 900 python tools/pmc_traffic.py --batch 32 --only front,wino6_l1_128_128 >> $OUT/pmc.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc_traffic_rows.txt $OUT/pmc_traffic_rows.txt
timeout 600 python bench.py --steps 20 --host-input > $OUT/bench_b1.json 2> $OUT/bench_b1.err
timeout 600 python bench.py --steps 10 --batch 8 --no-cpu-baseline > $OUT/bench_b8.json 2>/dev/null
timeout 600 python bench.py --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > $OUT/bench_b32.json 2>/dev/null
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o b1 -- python bench.py --steps 20 --no-cpu-baseline --no-kernel-timing > $OUT/bench_b1_under_rocprof.json 2>/dev/null
python tools/summarize_rocprof.py $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) 45 > $OUT/bench_b1_kernel_stats_summary.txt
python tools/step_profile.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 128 > $OUT/step_profile_b1.txt
timeout 600 python tools/bench_train.py > $OUT/bench_train.json 2>/dev/null
tail -3 $OUT/pmc.log; head -c 600 $OUT/bench_b1.json; echo; tail -c 300 $OUT/bench_train.json
