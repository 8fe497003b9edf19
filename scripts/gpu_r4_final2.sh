#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_r4_final2.sh 1"   (part 1: GPU suite, smoke, bench lines)
# gpurun --timeout 600 -- "bash scripts/gpu_r4_final2.sh 2"   (part 2: rocprofv3 kernel stats and PMC passes)
# Final pass of round 4's second half (two-wave fp16 kernels, guest blocks) on the round's last build.
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
if [ "$1" = "1" ]; then
  timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  grep -E "passed|failed|rc=" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_gpu.log | head; tail -2 $R/smoke.log
  timeout 300 python bench.py > $R/bench.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_f16x3_train.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --hidden 128 --layers 4 --precision f16x3_train > $R/bench_f16x3_train_4x128.log 2>&1
  timeout 100 python bench.py --mode eval --no-cpu-baseline --precision f16x3 > $R/bench_eval_f16x3.log 2>&1
  timeout 100 python bench.py --no-cpu-baseline --precision bf16x3_train > $R/bench_bf16x3_train.log 2>&1
  show bench bench_f16x3_train bench_f16x3_train_4x128 bench_eval_f16x3 bench_bf16x3_train
else
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_f16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_prof_f16.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  PMC_BENCH_ARGS="--precision f16x3_train" bash scripts/gpu_pmc.sh > $R/pmc_f16.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_f16x3_train.json; cp $R/pmc_summary.txt $R/pmc_summary_f16x3_train.txt
  bash scripts/gpu_pmc.sh > $R/pmc.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_8x256_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_8x256_4096.txt
  tail -12 $R/pmc_summary_f16x3_train.txt; tail -6 $R/pmc_summary_8x256_4096.txt
  find $R/prof_f16 -name "*kernel_stats.csv" | head -2
fi
