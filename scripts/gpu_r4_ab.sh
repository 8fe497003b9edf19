#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_r4_ab.sh 'libnerfhip.so libnerfhip_dense.so' 'f16x3_fwd_dgrad'"
# A/B of variant libraries (scripts/build_bf16_variant.sh) on the headline workload: each variant is copied over the box's copy of
# libnerfhip.so in turn, two rounds, kernel times from the bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
out=gpurun_out/r4_ab.txt; : > $out
for round in 1 2; do for lib in $1; do for p in $2; do
  src=nerf-pytorch_amd/$lib; [ "$lib" = "libnerfhip.so" ] && src=/tmp/libnerfhip_product.so
  cp $src nerf-pytorch_amd/libnerfhip.so
  timeout 200 python bench.py --no-cpu-baseline --precision $p 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', '$p', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['mlp_kernels'].items()})" >> $out
done; done; done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
cat $out
