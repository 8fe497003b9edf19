#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_r3_call1.sh"
# Round 3, first GPU pass: the GPU suite with the new parity records, smoke, the bench line (with shader-clock probe),
# the default 4x128 nets, the N > 1 paths on the one GPU of the box (self-launch, strong scaling, eval sharding),
# and the rocprofv3 kernel trace of the bench command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py > $R/bench.log 2>&1; echo "rc=$?" >> $R/bench.log
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline > $R/bench_4x128.log 2>&1
timeout 200 python bench.py --mode eval > $R/bench_eval.log 2>&1
export NERFHIP_BENCH_ONE_DEVICE=1
timeout 200 python bench.py --gpus 2 --steps 6 --warmup 2 > $R/dp2_weak.log 2>&1; echo "rc=$?" >> $R/dp2_weak.log
timeout 200 python bench.py --gpus 2 --steps 6 --warmup 2 --image 800 --global-rays 8192 > $R/dp2_strong.log 2>&1; echo "rc=$?" >> $R/dp2_strong.log
timeout 200 python bench.py --gpus 2 --mode eval --steps 2 --warmup 1 --gather > $R/dp2_eval.log 2>&1; echo "rc=$?" >> $R/dp2_eval.log
unset NERFHIP_BENCH_ONE_DEVICE
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "passed|failed|error" $R/pytest_gpu.log | tail -5; tail -2 $R/smoke.log; tail -2 $R/bench.log | cut -c1-3000; tail -1 $R/bench_4x128.log | cut -c1-1500; tail -1 $R/bench_eval.log | cut -c1-1200
for f in dp2_weak dp2_strong dp2_eval; do echo "== $f"; tail -2 $R/$f.log | cut -c1-700; done
