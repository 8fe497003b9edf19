#!/bin/bash
# gpurun --timeout 400 -- "bash scripts/gpu_r4_one_test.sh 'product noguest' 'tests/test_gpu_fullsize.py::test_lego_teacher_forced_fine_pass[f16x3_train]'"
# One GPU test on variant libraries (scripts/build_bf16_variant.sh), each copied over the box's libnerfhip.so in turn.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
for lib in $1; do
  src=nerf-pytorch_amd/libnerfhip_$lib.so; [ "$lib" = "product" ] && src=/tmp/libnerfhip_product.so
  cp $src nerf-pytorch_amd/libnerfhip.so
  timeout 300 python -m pytest "$2" -q -p no:cacheprovider -x > $R/one_test_$lib.log 2>&1
  echo "== $lib: $(grep -E 'passed|failed' $R/one_test_$lib.log | tail -1)"; grep -E "^E  " $R/one_test_$lib.log | head -3
  cp gpurun_out/parity_fullsize_lego_8x256_64+128_teacher_forced_f16x3_train.json gpurun_out/tf_$lib.json 2>/dev/null
done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
