#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_psnr_arms.sh ITERS 'SEEDS' 'REF_SEEDS' [extra psnr_arms.py args]"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ITERS=${1:-3000}; SEEDS=${2:-"1 2 3 4 5 6 7 8"}; REFS=${3:-"1 2 3 4"}; shift 3
O=gpurun_out/psnr_arms; mkdir -p $O
for s in $SEEDS; do
  arms="engine,engine_td,dropin"; for r in $REFS; do [ "$r" = "$s" ] && arms="$arms,ref"; done
  timeout 600 python scripts/psnr_arms.py $s $ITERS $O/seed$s.json --arms $arms "$@" > $O/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed$s.log | cut -c1-160)"
done
python scripts/psnr_stats.py $O > $O/stats.txt 2>&1; cat $O/stats.txt | head -60
