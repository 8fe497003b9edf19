#!/bin/bash
# gpurun --timeout 1800 -- "bash scripts/gpu_r4_psnr.sh 'SEEDS' [ITERS] [LR] [ARMS]"
# PSNR@iters at the metric's own geometry (8x256 students, 64 + 128, 4096 rays/iter): the reference's torch ops on this GPU (`ref`), the
# fp32 engine, and the engine on the f16x3 kernels -- same initial weights, views, pixels per seed; ONE learning rate for all arms.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; SEEDS=${1:-"1 2 3 4"}; ITERS=${2:-2000}; LR=${3:-1e-3}; ARMS=${4:-"engine,engine_f16tr,ref"}
O=gpurun_out/psnr_8x256; mkdir -p $O
for s in $SEEDS; do
  timeout 700 python scripts/psnr_arms.py $s $ITERS $O/seed$s.json --arms $ARMS --hidden 256 --layers 8 --lr $LR > $O/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed$s.log | cut -c1-200)"
done
python scripts/psnr_stats.py $O > $O/stats_partial.txt 2>&1; head -40 $O/stats_partial.txt
