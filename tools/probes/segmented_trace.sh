# queue-annotated one-step timeline of bench.py under "$@" (env assignments), e.g. M4D_SEGMENTED=1; output gpurun_out/seg_queue_trace_<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_seg_$TAG
env "$@" M4D_STAGGER_AUTOTUNE=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seg_$TAG -o t -- python bench.py --steps 20 --no-cpu-baseline --no-kernel-timing --no-configs2 > gpurun_out/seg_trace_bench_$TAG.json 2>/dev/null
TRACE=$(find /tmp/prof_seg_$TAG -name "*kernel_trace.csv" | head -1)
python tools/queue_trace.py "$TRACE" > gpurun_out/seg_queue_trace_$TAG.txt 2>/dev/null
python tools/probes/step_gaps.py "$TRACE" > gpurun_out/seg_step_gaps_$TAG.txt; cat gpurun_out/seg_step_gaps_$TAG.txt
