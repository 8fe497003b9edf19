#!/bin/bash
# gpurun -- "bash scripts/fern_ab.sh NAME [NAME ...]": the fern line of the fused backward on the A/B builds of scripts/build_r64_variant.sh
# (NAME "product": the tree's own build); MODES: the --compact values to run (default: fused_stash)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
for v in "$@"; do
  cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
  [ "$v" != product ] && cp nerf-pytorch_amd/libnerfhip_$v.so nerf-pytorch_amd/libnerfhip.so
  for m in ${MODES:-fused_stash}; do
  a="--compact $m --overlap 0"
  python bench.py --workload fern --no-cpu-baseline --no-labelled-lines $a 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $a |', d['ms_per_step'], 'unprofiled', d['unprofiled_rerun']['ms_per_step'], {k:(v['ms_per_step'], v['frac']) for k,v in r['mlp_kernels'].items()})"
  done
done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
