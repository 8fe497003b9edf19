cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
for p in f16x3_fwd f16x3_fwd_dgrad fp32+f16x3_train f16x3_train; do
  timeout 100 python bench.py --no-cpu-baseline --precision $p > $R/bench_$p.log 2>&1
  tail -1 $R/bench_$p.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$p', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['mlp_kernels'].items()})"
done
