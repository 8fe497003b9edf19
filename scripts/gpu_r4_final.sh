#!/bin/bash
# gpurun --timeout 2400 -- "bash scripts/gpu_r4_final.sh"
# Round 4 final pass on the round's last build: GPU suite, smoke, the bench lines (fp32 headline with its baselines and the labelled
# f16x3_train line, the other precisions, fern = BASELINE configs[3], 4x128, eval), rocprofv3 kernel stats, PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 400 python bench.py > $R/bench.log 2>&1
for p in f16x3_train f16x3_fwd_dgrad f16x3_fwd bf16x3_train fp32+bf16x3_train fp32+f16x3_train; do
  timeout 200 python bench.py --no-cpu-baseline --precision $p > $R/bench_$p.log 2>&1
done
timeout 300 python bench.py --workload fern > $R/bench_fern.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --hidden 128 --layers 4 > $R/bench_4x128.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --hidden 128 --layers 4 --precision f16x3_train > $R/bench_f16x3_train_4x128.log 2>&1
timeout 200 python bench.py --mode eval > $R/bench_eval.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision f16x3 > $R/bench_eval_f16x3.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision bf16x3 > $R/bench_eval_bf16x3.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_f16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_prof_f16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fern -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload fern > $R/bench_prof_fern.log 2>&1
cd $GRAFT_REPO_ROOT
bash scripts/gpu_pmc.sh > $R/pmc.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_8x256_4096.json; cp $R/pmc_summary.txt $R/pmc_summary_8x256_4096.txt
PMC_BENCH_ARGS="--precision f16x3_train" bash scripts/gpu_pmc.sh > $R/pmc_f16.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_f16x3_train.json; cp $R/pmc_summary.txt $R/pmc_summary_f16x3_train.txt
PMC_BENCH_ARGS="--workload fern" bash scripts/gpu_pmc.sh > $R/pmc_fern.log 2>&1; cp $R/pmc_summary.json $R/pmc_summary_fern.json; cp $R/pmc_summary.txt $R/pmc_summary_fern.txt
grep -E "passed|failed|rc=" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_gpu.log | head; tail -2 $R/smoke.log
for f in bench bench_f16x3_train bench_f16x3_fwd_dgrad bench_f16x3_fwd bench_bf16x3_train bench_fp32+bf16x3_train bench_fp32+f16x3_train bench_fern bench_4x128 bench_f16x3_train_4x128 bench_eval bench_eval_f16x3 bench_eval_bf16x3; do
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
tail -5 $R/pmc_summary_8x256_4096.txt
