#!/bin/bash
# Interleaved A/B of bench.py under different environments: tools/ab_bench.sh <repeats> "<env A>" "<env B>" ...
# AB_BENCH_ARGS="--batch 32" AB_STEPS="--steps 5 --warmup 2": other workloads / region lengths
# prints frames/s per run and the median per configuration (box-to-box and run-to-run noise is ~1 %).
R=${1:-3}; shift
cd "$(dirname "$0")/.."
declare -A vals
for ((i = 0; i < R; ++i)); do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py ${AB_STEPS:---steps 30 --warmup 3} --no-cpu-baseline --no-kernel-timing --no-configs2 $AB_BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    vals["$cfg"]+="$v "
  done
done
for cfg in "$@"; do
  python - "$cfg" ${vals["$cfg"]} <<'PY'
import sys, statistics
cfg, v = sys.argv[1], [float(x) for x in sys.argv[2:]]
print(f"{cfg:60s} median {statistics.median(v):8.1f}  runs {' '.join(f'{x:.1f}' for x in v)}")
PY
done
