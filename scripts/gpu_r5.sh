#!/bin/bash
# gpurun --timeout T -- "bash scripts/gpu_r5.sh PART [PART ...]"      (round 5: ONE parameterised script instead of a script per call)
#   tests   the GPU suite (pytest -m gpu) and smoke() on the library in the tree            -> gpurun_out/pytest_gpu.log, smoke.log
#   mocks   the loop mocks (scripts/f16w_loop_mock, scripts/ws_loop_mock: built in the container, they travel)  -> mocks.txt
#   bench   the driver's default line, then the labelled lines (f16x3_train, 4x128 fp32 / f16x3_train, fern, eval fp32 / f16x3)
#   prof    rocprofv3 --kernel-trace --stats of the default line and of the f16x3_train line, then the PMC passes of both
#   proxy   the one-GPU proxy lines of the 8-GPU strong-scaling shape;  fernfwd  fern on f16x3_fwd plans
#   soakshort  3000 iterations of the f16x3_train arm with the FILTERED gradient comparison every 250 (scripts/psnr_soak.py)
#   psnr8   8 seeds x 2000 iterations of the f16x3_train arm at 8x256 (scripts/psnr_arms.py; the fp32 arms: profiles/r04_psnr_8x256_runs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
show() {
  for f in "$@"; do
    echo "== $f"; tail -1 $R/$f.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print(d['value'], d['ms_per_step'], d.get('precision'), {k: (v['ms_per_step'], v['frac'], v['hbm_frac']) for k, v in (d['roofline'] or {}).get('mlp_kernels', {}).items()})
    if 'labelled_lines' in d: print('   labelled', {k: (v.get('value'), v.get('ms_per_step'), v.get('speedup_vs_pytorch_rocm_fwd_bwd')) for k, v in d['labelled_lines'].items()}, 'x torch', d.get('speedup_vs_pytorch_rocm_fwd_bwd'))
except Exception as e:
    print('unparsed', repr(e)[:200])
"
  done
}
for part in "$@"; do
case $part in
tests)
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  grep -E "passed|failed|rc=" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_gpu.log | head -20; tail -2 $R/smoke.log ;;
mocks)
  { echo "# scripts/f16w_loop_mock (the product's loop structure), then scripts/ws_loop_mock (weight-stationary), same box, back to back"
    timeout 120 scripts/f16w_loop_mock; timeout 120 scripts/ws_loop_mock; timeout 120 scripts/ws_loop_mock; } > $R/mocks.txt 2>&1
  cat $R/mocks.txt ;;
bench)
  timeout 400 python bench.py > $R/bench.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_f16x3_train.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --hidden 128 --layers 4 > $R/bench_4x128.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --hidden 128 --layers 4 --precision f16x3_train > $R/bench_f16x3_train_4x128.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --workload fern > $R/bench_fern_4x64.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --workload fern --precision f16x3_train > $R/bench_fern_4x64_f16x3_train.log 2>&1
  timeout 100 python bench.py --mode eval --no-cpu-baseline > $R/bench_eval.log 2>&1
  timeout 100 python bench.py --mode eval --no-cpu-baseline --precision f16x3 > $R/bench_eval_f16x3.log 2>&1
  show bench bench_f16x3_train bench_4x128 bench_f16x3_train_4x128 bench_fern_4x64 bench_fern_4x64_f16x3_train bench_eval bench_eval_f16x3 ;;
prof)
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_f16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_prof_f16.log 2>&1
  cd $GRAFT_REPO_ROOT
  bash scripts/gpu_pmc.sh > $R/pmc.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_8x256_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_8x256_4096.txt
  PMC_BENCH_ARGS="--precision f16x3_train" bash scripts/gpu_pmc.sh > $R/pmc_f16.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_f16x3_train.json; cp $R/pmc_summary.txt $R/pmc_summary_f16x3_train.txt
  tail -6 $R/pmc_summary_8x256_4096.txt; tail -12 $R/pmc_summary_f16x3_train.txt; find $R/prof $R/prof_f16 -name "*kernel_stats.csv" | head -2 ;;
psnr8)
  mkdir -p $R/psnr8
  for s in 1 2 3 4 5 6 7 8; do
    timeout 170 python scripts/psnr_arms.py $s 2000 $R/psnr8/seed$s.json --arms engine_f16tr --hidden 256 --layers 8 --lr 1e-3 > $R/psnr8/seed$s.log 2>&1
    echo "seed $s rc=$? $(grep 'engine_f16tr' $R/psnr8/seed$s.log | tail -1 | cut -c1-160)"
  done ;;
proxy)   # the per-GPU shape of BASELINE configs[2] (800x800, 8192 rays over 8 GPUs = 1024 per GPU) and the whole step, on one GPU
  for spec in "1024 fp32" "8192 fp32" "1024 f16x3_train" "8192 f16x3_train"; do
    set -- $spec
    timeout 120 python bench.py --no-cpu-baseline --image 800 --rays $1 --precision $2 > $R/bench_800_$1_$2.log 2>&1
  done
  show bench_800_1024_fp32 bench_800_8192_fp32 bench_800_1024_f16x3_train bench_800_8192_f16x3_train ;;
fernfwd)
  timeout 100 python bench.py --no-cpu-baseline --workload fern --precision f16x3_fwd > $R/bench_fern_4x64_f16x3_fwd.log 2>&1
  show bench_fern_4x64_f16x3_fwd ;;
soak)    # more seeds of the 20 000-iteration soak (scripts/gpu_r5_soak.sh ran seeds 1, 2): SOAK_SEEDS="3 4" bash scripts/gpu_r5.sh soak
  mkdir -p $R/soak
  for seed in ${SOAK_SEEDS:-3}; do
    timeout 1150 python scripts/psnr_soak.py $seed ${SOAK_ITERS:-20000} $R/soak/soak_seed$seed.json --arms engine_f16tr,engine > $R/soak/soak_seed$seed.log 2>&1
    echo "soak seed $seed rc=$?"; grep "val_psnr" $R/soak/soak_seed$seed.log | tail -2 | cut -c1-200
  done ;;
soakshort)
  mkdir -p $R/soak
  timeout 200 python scripts/psnr_soak.py 1 3000 $R/soak/soak_short_filtered_seed1.json --arms engine_f16tr --check 3000 --diag 250 > $R/soak/soak_short_filtered_seed1.log 2>&1
  echo "soakshort rc=$?"; grep diag $R/soak/soak_short_filtered_seed1.log | cut -c1-260 ;;
*) echo "unknown part $part" ;;
esac
done
