#!/bin/bash
# gpurun --timeout 660 -- "bash scripts/gpu_r4_w2.sh"
# First run of the two-waves-per-SIMD fp16 kernels (mlp_f16w.hip) on MI355X: the f16x3 part of the GPU suite on the product build, then
# the A/B against the one-wave kernels (libnerfhip_w1.so = scripts/build_bf16_variant.sh w1 "-DNHB_W2_DEFAULT=0 -DNHB_F16_ONE_WAVE"
# "plan mlp_f16") and the stores-first variant
# (libnerfhip_sf.so), one round each, most important lines first; every line is appended to gpurun_out/r4_w2.txt as it arrives.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
out=$R/r4_w2.txt; : > $out
line() {  # variant, bench args...
  lib=$1; shift
  src=nerf-pytorch_amd/libnerfhip_$lib.so; [ "$lib" = "product" ] && src=/tmp/libnerfhip_product.so
  cp $src nerf-pytorch_amd/libnerfhip.so
  timeout 150 python bench.py --no-cpu-baseline "$@" 2>$R/w2_err.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$lib', '$*', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in ((d.get('roofline') or {}).get('mlp_kernels') or {}).items()})
except Exception as e:
    print('$lib', '$*', 'unparsed', repr(e)[:120])" >> $out
  tail -3 $R/w2_err.log | cut -c1-300 >> $out
}
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
line product --precision f16x3_train
line w1 --precision f16x3_train
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so  # (the first run of this script tested the w1 library here by mistake)
timeout 420 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "f16x3 and not bf16" > $R/pytest_w2.log 2>&1; echo "pytest rc=$?" >> $R/pytest_w2.log
grep -E "passed|failed|rc=" $R/pytest_w2.log | tail -3 >> $out; grep -E "^FAILED|^ERROR" $R/pytest_w2.log | head >> $out
line sf --precision f16x3_train
line product --mode eval --precision f16x3
line w1 --mode eval --precision f16x3
line product --hidden 128 --layers 4 --precision f16x3_train
line w1 --hidden 128 --layers 4 --precision f16x3_train
line product --precision f16x3_train
line w1 --precision f16x3_train
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
cat $out
