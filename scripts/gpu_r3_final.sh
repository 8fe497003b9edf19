#!/bin/bash
# gpurun --timeout 1800 -- "bash scripts/gpu_r3_final.sh"
# Round 3 final pass: GPU suite, smoke, every bench line that DESIGN.md quotes, the N > 1 paths on the one GPU, rocprofv3
# kernel traces of the bench command (8x256 and 4x128) and the PMC passes (8x256 and 4x128).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py > $R/bench.log 2>&1; echo "rc=$?" >> $R/bench.log
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline > $R/bench_4x128.log 2>&1
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline --overlap 0 > $R/bench_4x128_single.log 2>&1
timeout 200 python bench.py --hidden 64 --layers 4 --no-cpu-baseline > $R/bench_4x64.log 2>&1
timeout 200 python bench.py --hidden 512 --layers 8 --no-cpu-baseline --steps 5 > $R/bench_8x512.log 2>&1
timeout 300 python bench.py --mode eval > $R/bench_eval.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision bf16x3 > $R/bench_eval_bf16x3.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --hidden 128 --layers 4 > $R/bench_eval_fp32_4x128.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --hidden 128 --layers 4 --precision bf16x3 > $R/bench_eval_bf16x3_4x128.log 2>&1
timeout 200 python scripts/bf16x3_timing.py > $R/bf16x3_timing.txt 2>&1
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_fwd > $R/bench_bf16x3_fwd.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_fwd --hidden 128 --layers 4 --overlap 0 > $R/bench_bf16x3_fwd_4x128.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_fwd_dgrad > $R/bench_bf16x3_fwd_dgrad.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_fwd_dgrad --hidden 128 --layers 4 --overlap 0 > $R/bench_bf16x3_fwd_dgrad_4x128.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_train > $R/bench_bf16x3_train.log 2>&1
for r in 2048 1024; do timeout 200 python bench.py --rays $r --no-cpu-baseline > $R/bench_rays$r.log 2>&1; done
export NERFHIP_BENCH_ONE_DEVICE=1
timeout 200 python bench.py --gpus 2 --steps 6 --warmup 2 > $R/dp2_weak.log 2>&1; echo "rc=$?" >> $R/dp2_weak.log
timeout 200 python bench.py --gpus 2 --steps 6 --warmup 2 --image 800 --global-rays 8192 > $R/dp2_strong.log 2>&1; echo "rc=$?" >> $R/dp2_strong.log
timeout 200 python bench.py --gpus 2 --mode eval --steps 2 --warmup 1 --gather > $R/dp2_eval.log 2>&1; echo "rc=$?" >> $R/dp2_eval.log
unset NERFHIP_BENCH_ONE_DEVICE
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof128 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --hidden 128 --layers 4 --overlap 0 > $R/bench_prof128.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_eval_bf16x3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode eval --precision bf16x3 --no-cpu-baseline > $R/bench_prof_eval_bf16x3.log 2>&1
cd $GRAFT_REPO_ROOT
bash scripts/gpu_bf16_pmc.sh > $R/bf16_pmc.log 2>&1
PMC_BENCH_ARGS="" bash scripts/gpu_pmc.sh > $R/pmc_8x256.log 2>&1
cp $R/pmc_summary.json $R/pmc_summary_8x256_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_8x256_4096.txt
PMC_BENCH_ARGS="--hidden 128 --layers 4" bash scripts/gpu_pmc.sh > $R/pmc_4x128.log 2>&1
cp $R/pmc_summary.json $R/pmc_summary_4x128_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_4x128_4096.txt
if [ -f nerf-pytorch_amd/libnerfhip_dbg.so ]; then
  timeout 200 python scripts/wgrad_timeline.py 786432 256 8 > $R/wgrad_timeline.txt 2>&1
  timeout 200 python scripts/wgrad_timeline.py 786432 128 4 > $R/wgrad_timeline_4x128.txt 2>&1
  timeout 200 python scripts/phase_timing.py > $R/phase_timing.txt 2>&1
fi
grep -E "passed|failed|error" $R/pytest_gpu.log | tail -3; tail -2 $R/smoke.log; tail -2 $R/bench.log | cut -c1-1500
for f in bench_4x128 bench_4x128_single bench_4x64 bench_8x512 bench_eval bench_eval_bf16x3 bench_eval_fp32_4x128 bench_eval_bf16x3_4x128 bench_bf16x3_fwd bench_bf16x3_fwd_4x128 bench_bf16x3_fwd_dgrad bench_bf16x3_fwd_dgrad_4x128 bench_bf16x3_train bench_rays2048 bench_rays1024 dp2_weak dp2_strong dp2_eval; do echo "== $f"; grep "^{" $R/$f.log | tail -1 | cut -c1-260; done
cat $R/pmc_summary_8x256_4096.txt $R/pmc_summary_4x128_4096.txt; ls $R/prof $R/prof128 | head
