#!/bin/bash
# gpurun -- "bash scripts/gpu_bf16_train_quick.sh": the whole-training-step parity test and the bf16x3_train bench line (8x256)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && R=gpurun_out
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "whole_training or bf16x3_train" 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline --precision bf16x3_train > $R/bench_bf16x3_train.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_bf16x3_train.log") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["final_loss"], {k: (v["ms_per_step"], v["frac"], v.get("peak")) for k, v in d["roofline"]["mlp_kernels"].items()})
PY
