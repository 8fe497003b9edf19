#!/bin/bash
# gpurun -- "bash scripts/gpu_bf16_fwd_train.sh": parity tests of the bf16x3 plans + the training bench with --precision bf16x3_fwd / bf16x3_fwd_dgrad
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && R=gpurun_out
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bf16" > $R/pytest_gpu_bf16.log 2>&1; tail -4 $R/pytest_gpu_bf16.log
for P in bf16x3_fwd bf16x3_fwd_dgrad bf16x3_train; do
  timeout 200 python bench.py --no-cpu-baseline --precision $P > $R/bench_$P.log 2>&1
  timeout 200 python bench.py --no-cpu-baseline --precision $P --hidden 128 --layers 4 --overlap 0 > $R/bench_${P}_4x128.log 2>&1
done
python - <<'PY'
import json
for f in ("bench_bf16x3_fwd", "bench_bf16x3_fwd_4x128", "bench_bf16x3_fwd_dgrad", "bench_bf16x3_fwd_dgrad_4x128", "bench_bf16x3_train", "bench_bf16x3_train_4x128"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.log" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["final_loss"], {k: (v["ms_per_step"], v["frac"], v.get("peak")) for k, v in d["roofline"]["mlp_kernels"].items()})
    except Exception as e:
        print(f, "failed", repr(e)[:200])
PY
