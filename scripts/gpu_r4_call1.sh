#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_r4_call1.sh"
# Round 4, first pass: the whole GPU suite (now with the f16x3 plans under the fp32 kernels' assertions), smoke, and the bench
# lines of the new precisions next to the fp32 headline and round 3's bf16x3_train.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py > $R/bench.log 2>&1
for p in f16x3_train f16x3_fwd_dgrad f16x3_fwd bf16x3_train fp32+f16x3_train; do
  timeout 200 python bench.py --no-cpu-baseline --precision $p > $R/bench_$p.log 2>&1
done
timeout 200 python bench.py --no-cpu-baseline --hidden 128 --layers 4 --precision f16x3_train > $R/bench_f16x3_train_4x128.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --hidden 128 --layers 4 > $R/bench_4x128.log 2>&1
timeout 300 python bench.py --workload fern > $R/bench_fern.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision f16x3 > $R/bench_eval_f16x3.log 2>&1
timeout 200 python bench.py --mode eval --no-cpu-baseline > $R/bench_eval.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_f16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --precision f16x3_train > $R/bench_prof_f16.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "passed|failed|rc=" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_gpu.log | head -20; tail -2 $R/smoke.log
for f in bench bench_f16x3_train bench_f16x3_fwd_dgrad bench_f16x3_fwd bench_bf16x3_train bench_fp32+f16x3_train bench_f16x3_train_4x128 bench_4x128 bench_fern bench_eval_f16x3 bench_eval; do
  echo "== $f"; tail -1 $R/$f.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print(d['value'], d['ms_per_step'], d.get('precision'), {k: (v['ms_per_step'], v['frac']) for k, v in (d['roofline'] or {}).get('mlp_kernels', {}).items()})
except Exception as e:
    print('unparsed', repr(e)[:200])
"
done
