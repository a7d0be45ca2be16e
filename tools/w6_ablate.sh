#!/bin/bash
# Timing ablations of the bf16-split Winograd kernel's K loop (m4d_wino6.hip, -DM4D_W6_ABL=bits: 1 no per-position barrier,
# 2 no fragment DMA, 4 no raw-halo DMA, 8 no fragment LDS reads, 16 no A-operand generation, 32 no DMA waits; results are
# wrong by construction).  Expects build_tmp/abl/lib_<bits>.so (the library linked with the kernel built that way; lib_0 =
# the product) and swaps them in one after the other.  usage (GPU box, repo root): bash tools/w6_ablate.sh
cd "$(dirname "$0")/.."
cp m4depth_amd/libm4depth_hip.so /tmp/lib_product.so
for ab in 0 1 2 4 6 8 16 32 63 0; do
  [ -f build_tmp/abl/lib_$ab.so ] || continue
  cp build_tmp/abl/lib_$ab.so m4depth_amd/libm4depth_hip.so
  for args in "--cin 128 --cout 128 --h 192 --w 640" "--cin 128 --cout 128 --h 192 --w 640 --batch 8"; do
    echo -n "ablate $ab: "; timeout 120 python tools/bench_conv_one.py $args --winograd 6 --iters 50 2>&1 | grep "^conv"
  done
done
cp /tmp/lib_product.so m4depth_amd/libm4depth_hip.so
