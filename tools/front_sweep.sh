mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
run() {  # label, env...
  label=$1; shift
  rm -rf /tmp/prof_$label
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -o x -- python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing > /tmp/b_$label.json 2>/dev/null
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  echo "== $label fps $(python -c "import json;print(json.load(open('/tmp/b_$label.json'))['value'])")"
  python tools/summarize_rocprof.py $f 60 | grep level_front | cut -c1-60,112-150
}
run base A=1
run l2v1 M4D_FRONT_L2=1
run l2v2 M4D_FRONT_L2=2
run l2v3 M4D_FRONT_L2=3
run l2v4 M4D_FRONT_L2=4
run l3v1 M4D_FRONT_L3=1
run l3v2 M4D_FRONT_L3=2
run l3v3 M4D_FRONT_L3=3
run l3v4 M4D_FRONT_L3=4
run l3v5 M4D_FRONT_L3=5
run l1v1 M4D_FRONT_L1=1
run l1v3 M4D_FRONT_L1=3
