#!/bin/bash
# frames/s of the batch-1 bench under one environment knob at a time (A/B on one box; each run ~8 s)
run() { echo "$* -> $(env "$@" python bench.py --steps 20 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])")"; }
run A=0
for kv in "$@"; do run $kv; done
run A=0
