#!/bin/bash
# gpurun --timeout 400 -- "bash scripts/gpu_r4_w2_diag.sh 'product noepi nostore ...' '--precision f16x3_train'"
# What bounds the two-wave fp16 kernels: diagnostic builds (scripts/build_bf16_variant.sh NAME -DNHW_EXP_* mlp_f16w; wrong results) on
# the headline workload, kernel times from the bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
out=$R/r4_w2_diag.txt; : > $out
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
for lib in $1; do
  src=nerf-pytorch_amd/libnerfhip_$lib.so; [ "$lib" = "product" ] && src=/tmp/libnerfhip_product.so
  cp $src nerf-pytorch_amd/libnerfhip.so
  timeout 120 python bench.py --no-cpu-baseline $2 2>$R/w2_err.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in ((d.get('roofline') or {}).get('mlp_kernels') or {}).items()})
    km = (d.get('roofline') or {}).get('kernel_ms_per_step') or d.get('kernel_ms_per_step') or {}
    print('   ', {k: v for k, v in km.items() if 'wgrad' in k})
except Exception as e:
    print('$lib', 'unparsed', repr(e)[:120])" >> $out
done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
cat $out
