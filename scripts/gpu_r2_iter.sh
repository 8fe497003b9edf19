#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_r2_iter.sh [tag] [pytest -k expr]"   (tuning loop: quick parity subset + bench lines)
cd "$GRAFT_REPO_ROOT"; TAG=${1:-iter}; K=${2:-"mlp or e2e or northstar or linearity or lego"}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -3
rm -f $O/bench.jsonl
for extra in "--overlap 0" "--overlap 1" "--overlap 0 --hidden 128 --layers 4" "--overlap 1 --hidden 128 --layers 4"; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $extra 2>> $O/bench.err >> $O/bench.jsonl
done
python scripts/bench_summary.py "8x256 overlap0" "8x256 overlap1" "4x128 overlap0" "4x128 overlap1" < $O/bench.jsonl
