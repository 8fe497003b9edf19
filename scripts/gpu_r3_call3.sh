#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_r3_call3.sh"
# Round 3, third GPU pass: whole GPU suite (512-wide instances, ray gradients), an 8x512 bench line, and the PSNR arms at
# the metric's student size (8x256): 4 seeds x {engine, dropin} x 2000 iterations.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 200 python bench.py --hidden 512 --layers 8 --no-cpu-baseline --steps 5 > $R/bench_8x512.log 2>&1
grep -E "passed|failed|error" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^E  " $R/pytest_gpu.log | head -20; tail -1 $R/bench_8x512.log | cut -c1-900
O=gpurun_out/psnr_arms_8x256; mkdir -p $O
for s in 1 2 3 4; do
  timeout 600 python scripts/psnr_arms.py $s 2000 $O/seed$s.json --arms engine,dropin --hidden 256 --layers 8 > $O/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed$s.log | cut -c1-160)"
done
python scripts/psnr_stats.py $O > $O/stats.txt 2>&1; head -30 $O/stats.txt
