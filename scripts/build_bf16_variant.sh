#!/bin/bash
# scripts/build_bf16_variant.sh NAME "-DFLAGS" -> nerf-pytorch_amd/libnerfhip_NAME.so: mlp_bf16.hip recompiled with FLAGS, linked
# with the product build's other objects (make lib first).  A/B and diagnostic builds only; never loaded by the package.
set -e
cd "$(dirname "$0")/../nerf-pytorch_amd/csrc"
mkdir -p build_var_$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000 $2 -c mlp_bf16.hip -o build_var_$1/mlp_bf16.o
OBJS=""
for f in elementwise dataio render sample mlp mlp16 mlp16_w512 mlp16_ext wgrad fused plan; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerfhip_$1.so $OBJS build_var_$1/mlp_bf16.o
rm -rf build_var_$1
echo built libnerfhip_$1.so
