#!/bin/bash
# Round evidence, every artefact checked non-empty (the script exits non-zero naming the first empty one):
#   bench lines (batch 1 with the PCIe-inclusive rate, 8, 32); rocprofv3 kernel-trace + stats of the batch-1 and batch-32
#   bench -> per-kernel summary, per (kernel, grid) launch durations (level-1 front / tail are their own grids), step profile;
#   kernel-trace durations of the roofline layer alone (level-1 refiner 128->128, batch 1 and 32); PMC passes (counters
#   only, one rocprofv3 run per counter group): HBM traffic (profiles/pmc_traffic.json) and the matrix-core / issue counters
#   of the roofline kernel.
# usage (GPU box, repo root): bash tools/collect_round_profiles.sh r04   -> gpurun_out/<tag>/ ; copy what is cited into profiles/
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
FAILED=""
need() { for f in "$@"; do if [ ! -s "$f" ]; then echo "[collect] EMPTY OR MISSING: $f" >&2; FAILED="$FAILED $f"; fi; done; }

# PMC first: HBM traffic (stamped json: bench.py refuses entries whose kernel sources changed since) and the roofline
# kernel's issue / matrix-core counters
if [ "${SKIP_PMC:-0}" != 1 ]; then   # (SKIP_PMC=1: the counters of this checkout are already in profiles/pmc_traffic.json -- stamped by source hash)
timeout 1800 python tools/pmc_traffic.py --batch 1 > $OUT/pmc.log 2>&1
timeout 1500 python tools/pmc_traffic.py --batch 32 --only front,wino6_l1_128_128,front_l4,dscv_l4,sncv_l4 >> $OUT/pmc.log 2>&1
timeout 900 python tools/pmc_traffic.py --config4 >> $OUT/pmc.log 2>&1          # BASELINE configs[4] geometry (round 6): front + the roofline layer
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp gpurun_out/pmc_traffic_rows.txt $OUT/pmc_traffic_rows.txt
need $OUT/pmc_traffic.json $OUT/pmc_traffic_rows.txt
timeout 1200 bash tools/pmc_wino6.sh > /dev/null 2>&1; cp gpurun_out/pmc/wino6.txt $OUT/wino6_pmc.txt; need $OUT/wino6_pmc.txt
fi


timeout 600 python bench.py --steps 20 --host-input > $OUT/bench_b1.json 2> $OUT/bench_b1.err;                      need $OUT/bench_b1.json
timeout 600 python bench.py --steps 10 --batch 8 --no-cpu-baseline > $OUT/bench_b8.json 2> $OUT/bench_b8.err;       need $OUT/bench_b8.json
timeout 600 python bench.py --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; need $OUT/bench_b32.json

for B in 1 32; do
  rm -rf /tmp/prof_${TAG}_b$B
  STEPS=20; [ $B = 32 ] && STEPS=4
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_b$B -o t -- \
      python bench.py --steps $STEPS --batch $B --no-cpu-baseline --no-kernel-timing --no-configs2 > $OUT/bench_b${B}_under_rocprof.json 2> $OUT/bench_b${B}_under_rocprof.err
  STATS=$(find /tmp/prof_${TAG}_b$B -name "*kernel_stats.csv" | head -1); TRACE=$(find /tmp/prof_${TAG}_b$B -name "*kernel_trace.csv" | head -1)
  if [ -z "$STATS" ] || [ -z "$TRACE" ]; then echo "[collect] rocprofv3 wrote no kernel stats / trace for batch $B" >&2; FAILED="$FAILED rocprof_b$B"; continue; fi
  python tools/summarize_rocprof.py "$STATS" 45 > $OUT/bench_b${B}_kernel_stats_summary.txt
  python tools/trace_table.py "$TRACE" 0.3 > $OUT/bench_b${B}_launch_durations_by_grid.txt
  [ $B = 1 ] && python tools/step_profile.py "$TRACE" 128 1 > $OUT/step_profile_b1.txt
  [ $B = 1 ] && python tools/chain_trace.py "$TRACE" 4000 > $OUT/step_timeline_b1.txt
  need $OUT/bench_b${B}_kernel_stats_summary.txt $OUT/bench_b${B}_launch_durations_by_grid.txt
done
need $OUT/step_profile_b1.txt $OUT/step_timeline_b1.txt

# the roofline layer alone: every launch of the trace is the level-1 refiner 128->128 layer
for B in 1 32; do
  rm -rf /tmp/prof_${TAG}_layer_b$B
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_layer_b$B -o t -- \
      python tools/bench_conv_one.py --winograd 6 --iters 20 --batch $B > $OUT/roofline_layer_b${B}_hip_events.txt 2>&1
  TRACE=$(find /tmp/prof_${TAG}_layer_b$B -name "*kernel_trace.csv" | head -1)
  if [ -n "$TRACE" ]; then python tools/trace_table.py "$TRACE" 0.0 wino6 > $OUT/roofline_layer_b${B}_rocprof_durations.txt; fi
  need $OUT/roofline_layer_b${B}_rocprof_durations.txt $OUT/roofline_layer_b${B}_hip_events.txt
done

# the level tail: fp32-MFMA kernel (m4d_tail.hip) against the bf16-split persistent kernel (m4d_tail6.hip), graph-replayed launches
{ for SZ in "" "--batch 32" "--h 96 --w 320" "--h 48 --w 160" "--h 12 --w 40"; do
    timeout 120 python tools/bench_tail.py --iters 100 $SZ 2>&1 | grep "^refiner"; timeout 120 python tools/bench_tail.py --iters 100 $SZ --split 2>&1 | grep "^refiner"; done; } > $OUT/refiner_tail_split_vs_fp32.txt
need $OUT/refiner_tail_split_vs_fp32.txt

# the persistent bf16-split Winograd kernel against the one-unit kernel: per-layer times and bit equality (batch 1 and 8)
{ timeout 300 python tools/bench_wino6p.py; timeout 300 python tools/bench_wino6p.py --batch 8 --iters 5; } > $OUT/wino6_persistent_vs_one_unit.txt 2>&1; need $OUT/wino6_persistent_vs_one_unit.txt
# queue-annotated timeline of one step (the hipGraph executor's layout)
rm -rf /tmp/prof_${TAG}_q
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_q -o t -- python bench.py --steps 20 --no-cpu-baseline --no-kernel-timing --no-configs2 > /dev/null 2>&1
TRACE=$(find /tmp/prof_${TAG}_q -name "*kernel_trace.csv" | head -1)
[ -n "$TRACE" ] && python tools/queue_trace.py "$TRACE" > $OUT/queue_trace_b1_graph.txt 2> /dev/null
need $OUT/queue_trace_b1_graph.txt

# the captured graph's kernel nodes at batch 1 and 32 (every node one of libm4depth_hip.so's: the tool exits non-zero otherwise)
for B in 1 32; do
  timeout 600 python tools/dump_graph_nodes.py --batch $B --out $OUT/hipgraph_nodes_b$B.dot > $OUT/hipgraph_nodes_b$B.txt 2>&1 || FAILED="$FAILED graph_nodes_b$B"
  need $OUT/hipgraph_nodes_b$B.dot $OUT/hipgraph_nodes_b$B.txt
done
# the latency-first small-map convolution: every configuration per coarse-level layer, beside conv3x3_small6 (network_ops.lat_config's table)
timeout 900 python tools/bench_lat_convs.py sweep > $OUT/lat_conv_sweep.txt 2>&1; need $OUT/lat_conv_sweep.txt
# BASELINE configs[4] (768x2560, search ranges 6 / 6): fused fronts of levels 1-4 (m4d_level_front_r)
timeout 600 python bench.py --height 768 --width 2560 --dscv-range 6 --sncv-range 6 --steps 10 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err; need $OUT/bench_config4.json

timeout 600 python tools/bench_train.py > $OUT/bench_train.json 2>/dev/null; need $OUT/bench_train.json
head -c 400 $OUT/bench_b1.json; echo
if [ -n "$FAILED" ]; then echo "[collect] FAILED artefacts:$FAILED" >&2; exit 1; fi
echo "[collect] all artefacts non-empty in $OUT"
