#!/bin/bash
# gpurun --timeout 1200 -- "bash scripts/gpu_r3_call4.sh"
# Round 3, fourth GPU pass: the three tests that failed in pass 3 (robust comparisons now), the split-bf16 inner-loop
# mock, A/B of the 4-wave / two-workgroups-per-CU shape for the 256-wide forward / data-gradient kernels, PMC passes of
# the headline command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 500 -p no:cacheprovider -k "512 or ray or backward or padded or engine" > $R/pytest_gpu_sub.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu_sub.log
grep -E "passed|failed|error" $R/pytest_gpu_sub.log | tail -3; grep -E "^FAILED|^E  " $R/pytest_gpu_sub.log | head -10
timeout 120 scripts/split_bf16_mock > $R/split_bf16_mock.txt 2>&1; cat $R/split_bf16_mock.txt
bash scripts/gpu_ab.sh "base wg4" 2>&1 | tee $R/ab/ab_summary.txt
bash scripts/gpu_ab.sh "base" --hidden 128 --layers 4 2>&1 | tail -3
bash scripts/gpu_ab.sh "base" --hidden 64 --layers 4 2>&1 | tail -3
PMC_BENCH_ARGS="" bash scripts/gpu_pmc.sh > $R/pmc_8x256.log 2>&1
cp $R/pmc_summary.json $R/pmc_summary_8x256_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_8x256_4096.txt; cat $R/pmc_summary_8x256_4096.txt
