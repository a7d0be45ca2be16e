#!/bin/bash
# per-launch time of the bf16-split Winograd kernel with parts of its K loop removed (build with make W6FLAGS=-DM4D_W6_ABLATIONS)
for abl in 0 1 2 3 4 12 28 29 31; do
  echo "ABLATE=$abl: $(M4D_WINO6_ABLATE=$abl python tools/bench_wino6.py --check 0 2>/dev/null | grep -E '192x640 128->128|96x320 128->128' | sed 's/fp32.*bf16x6/bf16x6/' | tr '\n' ' ')"
done
