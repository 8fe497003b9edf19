#!/bin/bash
# scripts/build_bf16_variant.sh NAME "-DFLAGS" ["FILE ..."] -> nerf-pytorch_amd/libnerfhip_NAME.so: the named sources (default mlp_bf16;
# stems of .hip files, or plan for plan.cpp) recompiled with FLAGS, linked with the product build's other objects (make lib first).
# A/B and diagnostic builds only; never loaded by the package.
set -e
cd "$(dirname "$0")/../nerf-pytorch_amd/csrc"
FILES=${3:-mlp_bf16}
mkdir -p build_var_$1
for F in $FILES; do
  SRC=$F.hip; [ "$F" = "plan" ] && SRC=plan.cpp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000 $2 -c $SRC -o build_var_$1/$F.o
done
OBJS=""
for f in elementwise dataio render sample mlp mlp16 mlp16_w512 mlp16_ext mlp_bf16 mlp_f16 mlp_f16w wgrad wgrad_bf16 wgrad_f16 fused plan; do
  if [ -f build_var_$1/$f.o ]; then OBJS="$OBJS build_var_$1/$f.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerfhip_$1.so $OBJS
rm -rf build_var_$1
echo built libnerfhip_$1.so
