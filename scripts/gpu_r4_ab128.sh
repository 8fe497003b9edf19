#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_r4_ab128.sh 'libnerfhip.so libnerfhip_big128.so' 'f16x3_train' [pytest -k expression]"
# A/B of variant libraries on the reference's default 4x128 nets: each variant is copied over the box's copy of
# libnerfhip.so in turn, two rounds; optionally the GPU parity tests selected by the expression run on the LAST variant.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
out=gpurun_out/r4_ab128.txt; : > $out
for round in 1 2; do for lib in $1; do for p in $2; do
  src=nerf-pytorch_amd/$lib; [ "$lib" = "libnerfhip.so" ] && src=/tmp/libnerfhip_product.so
  cp $src nerf-pytorch_amd/libnerfhip.so
  timeout 200 python bench.py --no-cpu-baseline --hidden 128 --layers 4 --precision $p 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', '$p', d['value'], d['ms_per_step'], (d.get('unprofiled_rerun') or {}).get('ms_per_step'), d['roofline']['kernel_ms_per_step'])" >> $out
done; done; done
if [ -n "$3" ]; then timeout 500 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "$3" 2>&1 | tail -5 >> $out; fi
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
cat $out
