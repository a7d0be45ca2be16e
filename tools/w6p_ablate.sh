#!/bin/bash
# Timing ablations of the persistent bf16-split Winograd kernel (m4d_wino6p.hip): rebuilds the library with -DM4D_W6P_ABL=bits
# (results are then WRONG), times the level-1 layers at batch 8, restores the product build.  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
for abl in 0 16 18; do  # 1 = no epilogue passes (accumulators kept alive), 2 = no global stores, 8 = no waits in positions 0-1 of a unit, 16 = start skew
  rm -f m4depth_amd/csrc/build/m4d_wino6p.o
  make -C m4depth_amd/csrc W6FLAGS=-DM4D_W6P_ABL=$abl > /dev/null 2>&1
  echo "== M4D_W6P_ABL=$abl"
  python tools/bench_wino6p.py --batch ${W6P_BATCH:-8} --iters 5 2>/dev/null | grep "192x640 128->128\|192x640  64->128"
done
rm -f m4depth_amd/csrc/build/m4d_wino6p.o
make -C m4depth_amd/csrc > /dev/null 2>&1
