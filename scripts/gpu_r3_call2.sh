#!/bin/bash
# gpurun --timeout 2000 -- "bash scripts/gpu_r3_call2.sh"
# Round 3, second GPU pass: the 64-wide kernel instances (parity + bench line), PMC passes for the 4x128 nets, and the
# PSNR arms experiment (8 seeds x {engine, engine on torch's draws, drop-in} + 4 seeds of the reference path).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 500 -p no:cacheprovider -k "64 or external_draws or eval_800 or padded" > $R/pytest_gpu_64.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu_64.log
timeout 200 python bench.py --hidden 64 --layers 4 --no-cpu-baseline > $R/bench_4x64.log 2>&1
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline --overlap 0 > $R/bench_4x128_single.log 2>&1
PMC_BENCH_ARGS="--hidden 128 --layers 4" bash scripts/gpu_pmc.sh > $R/pmc_4x128.log 2>&1
cp $R/pmc_summary.json $R/pmc_summary_4x128_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_4x128_4096.txt
grep -E "passed|failed|error" $R/pytest_gpu_64.log | tail -3; tail -1 $R/bench_4x64.log | cut -c1-1800; tail -12 $R/pmc_summary_4x128_4096.txt
bash scripts/gpu_psnr_arms.sh 3000 "1 2 3 4 5 6 7 8" "1 2 3 4"
