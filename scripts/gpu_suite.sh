#!/bin/bash
# Runs on the GPU box via:  gpurun --timeout 1500 -- "bash scripts/gpu_suite.sh"
# gpu test-suite, smoke(), bench (default + the 4x128 nets the reference's scripts really build), rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py > $R/bench.log 2>&1
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline > $R/bench_4x128.log 2>&1
for r in 2048 1024; do timeout 200 python bench.py --rays $r --no-cpu-baseline > $R/bench_rays$r.log 2>&1; done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/eval_bench.py > $R/eval.log 2>&1
grep -E "passed|failed" $R/pytest_gpu.log | tail -2; tail -2 $R/smoke.log; tail -1 $R/bench.log | cut -c1-2600; tail -1 $R/bench_4x128.log | cut -c1-700; tail -1 $R/bench_rays2048.log | cut -c1-300; tail -1 $R/bench_rays1024.log | cut -c1-300; tail -1 $R/eval.log | cut -c1-500; ls $R/prof | head
