#!/bin/bash
# Timing ablations of the bf16-split Winograd kernel's K loop (m4d_wino6.hip, -DM4D_W6_ABL=bits: 1 no per-position barrier,
# 2 no fragment DMA, 4 no raw-halo DMA, 8 no fragment LDS reads, 16 no A-operand generation, 32 no DMA waits; results are
# wrong by construction).
#   bash tools/w6_ablate.sh build   (anywhere hipcc runs, after `make -C m4depth_amd/csrc`): links one library per variant
#                                   into build_tmp/abl/lib_<bits>.so (lib_0 = the product)
#   bash tools/w6_ablate.sh         (GPU box, repo root): swaps them in one after the other and times the level-1 128->128 layer
cd "$(dirname "$0")/.."
VARIANTS="1 2 4 6 8 16 32 63"
if [ "${1:-}" = build ]; then
  mkdir -p build_tmp/abl
  cp m4depth_amd/libm4depth_hip.so build_tmp/abl/lib_0.so
  OBJS=$(ls m4depth_amd/csrc/build/*.o | grep -v "m4d_wino6.o")
  for ab in $VARIANTS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function -Wno-pass-failed \
          -fno-slp-vectorize -DM4D_W6_ABL=$ab -c m4depth_amd/csrc/m4d_wino6.hip -o build_tmp/abl/w6_$ab.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/abl/lib_$ab.so $OBJS build_tmp/abl/w6_$ab.o ) &
  done
  wait
  ls -la build_tmp/abl/*.so
  exit 0
fi
cp m4depth_amd/libm4depth_hip.so /tmp/lib_product.so
for ab in 0 $VARIANTS 0; do
  [ -f build_tmp/abl/lib_$ab.so ] || continue
  cp build_tmp/abl/lib_$ab.so m4depth_amd/libm4depth_hip.so
  for args in "--cin 128 --cout 128 --h 192 --w 640" "--cin 128 --cout 128 --h 192 --w 640 --batch 8"; do
    echo -n "ablate $ab: "; timeout 120 python tools/bench_conv_one.py $args --winograd 6 --iters 50 2>&1 | grep "^conv"
  done
done
cp /tmp/lib_product.so m4depth_amd/libm4depth_hip.so
