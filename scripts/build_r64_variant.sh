#!/bin/bash
# bash scripts/build_r64_variant.sh NAME "DEFS": nerf-pytorch_amd/libnerfhip_NAME.so = the product objects with mlp64r.hip recompiled
# with DEFS (A/B builds of the fused 64-wide backward only; never loaded by the package)
set -e
cd "$(dirname "$0")/../nerf-pytorch_amd/csrc"
mkdir -p build_ab
F="--offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000"
/opt/rocm/bin/hipcc $F $2 -c mlp64r.hip -o build_ab/mlp64r_$1.o &
/opt/rocm/bin/hipcc $F $2 -c plan.cpp -o build_ab/plan_$1.o &
/opt/rocm/bin/hipcc $F $2 -c mlp.hip -o build_ab/mlp_$1.o &
wait
OBJS=$(ls build/*.o | grep -v "emu_" | grep -v "/mlp64r.o" | grep -v "/plan.o" | grep -v "/mlp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerfhip_$1.so $OBJS build_ab/mlp64r_$1.o build_ab/plan_$1.o build_ab/mlp_$1.o
ls -la ../libnerfhip_$1.so
