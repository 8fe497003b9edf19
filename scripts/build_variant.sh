#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAGS"  ->  nerf-pytorch_amd/libnerfhip_NAME.so  (A/B tuning builds; never loaded by the package)
set -e
cd "$(dirname "$0")/../nerf-pytorch_amd/csrc"; NAME=$1; FLAGS=$2; B=build_$NAME; mkdir -p $B
F="--offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000 $FLAGS"
for f in elementwise.hip dataio.hip render.hip sample.hip mlp.hip mlp16.hip mlp16_w512.hip mlp16_ext.hip mlp_bf16.hip wgrad.hip wgrad_bf16.hip fused.hip plan.cpp; do
  X=""; [ "$f" = wgrad.hip ] && X="-fno-slp-vectorize"   # (as the Makefile does)
  /opt/rocm/bin/hipcc $F $X -c $f -o $B/${f%.*}.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerfhip_$NAME.so $B/*.o
echo built ../libnerfhip_$NAME.so
