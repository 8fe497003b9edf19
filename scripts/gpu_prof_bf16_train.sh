#!/bin/bash
# gpurun -- "bash scripts/gpu_prof_bf16_train.sh": rocprofv3 kernel trace of bench.py --precision bf16x3_train
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT/gpurun_out && mkdir -p $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_bf16x3_train -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --precision bf16x3_train > $R/bench_prof_bf16x3_train.log 2>&1
grep "^{" $R/bench_prof_bf16x3_train.log | cut -c1-160; head -8 $R/prof_bf16x3_train/bench_kernel_stats.csv | cut -c1-160
