#!/bin/bash
# gpurun -- "bash scripts/narrow_ab.sh NAME [NAME ...]": the 4x128 line (what train_nerf.py:117-134 builds) on A/B builds of mlp16.hip /
# wgrad.hip (NAME "product": the tree's own build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
for v in "$@"; do
  cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
  [ "$v" != product ] && cp nerf-pytorch_amd/libnerfhip_$v.so nerf-pytorch_amd/libnerfhip.so
  for a in "--hidden 128 --layers 4 --overlap 0" ${EXTRA:+"$EXTRA"}; do
  python bench.py --no-cpu-baseline --no-labelled-lines $a 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $a |', d['ms_per_step'], 'unprofiled', d['unprofiled_rerun']['ms_per_step'], {k:(v['ms_per_step'], v['frac']) for k,v in r['mlp_kernels'].items()})"
  done
done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
