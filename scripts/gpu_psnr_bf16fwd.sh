#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_psnr_bf16fwd.sh ITERS 'SEEDS'": the engine arm against the same arm with the forward
# passes on the split-bf16 kernel (NERFHIP_PRECISION_BF16X3_FWD), paired per seed -- the acceptance test of that arithmetic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ITERS=${1:-3000}; SEEDS=${2:-"1 2 3 4 5 6 7 8"}; ARMS=${3:-engine,engine_bf16fwd}; OUT=${4:-psnr_bf16fwd}; EXTRA=${5:-}
O=gpurun_out/$OUT; mkdir -p $O
for s in $SEEDS; do
  timeout 400 python scripts/psnr_arms.py $s $ITERS $O/seed$s.json --arms $ARMS $EXTRA > $O/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed$s.log | cut -c1-200)"
done
python scripts/psnr_stats.py $O > $O/stats.txt 2>&1; cat $O/stats.txt | head -70
