#!/bin/bash
# gpurun --timeout 2400 -- "bash scripts/gpu_psnr400.sh ITERS 'SEEDS' 'ARMS'"   e.g.  5000 '0 1 2' 'engine dropin ref'
cd "$GRAFT_REPO_ROOT"; ITERS=${1:-5000}; SEEDS=${2:-"0 1 2"}; ARMS=${3:-"engine dropin ref"}; O=gpurun_out/psnr400; mkdir -p $O
for s in $SEEDS; do for arm in $ARMS; do
  timeout 900 python scripts/psnr400.py $arm $s $ITERS $O/${arm}_seed$s.json > $O/${arm}_seed$s.log 2>&1; echo "$arm seed $s rc=$? $(tail -1 $O/${arm}_seed$s.log | cut -c1-200)"
done; done
