#!/bin/bash
# gpurun -- "bash scripts/gpu_r3_last_check.sh": the GPU suite and the bf16x3_train bench line on the round's last build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && R=gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 500 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log; tail -3 $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log; tail -2 $R/smoke.log
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_train > $R/bench_bf16x3_train.log 2>&1; grep "^{" $R/bench_bf16x3_train.log | cut -c1-200
