#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_r4_short.sh [pytest -k expression]"
# The short-step lines (fern = BASELINE configs[3]; the reference's default 4x128 nets, fp32 and f16x3_train), two rounds, with the
# un-profiled re-run of each; optionally the GPU parity tests selected by the expression.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r4_short.txt; : > $out
for round in 1 2; do
  for a in "--workload fern" "--hidden 128 --layers 4" "--hidden 128 --layers 4 --precision f16x3_train" "--hidden 128 --layers 4 --precision bf16x3_train"; do
    timeout 200 python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 > /tmp/line.json
    n=$(echo "$a" | tr -d ' -'); [ $round = 1 ] && cp /tmp/line.json gpurun_out/short_line_$n.json
    python -c "
import sys, json
d = json.loads(open('/tmp/line.json').read()); print('$a', d['value'], d['ms_per_step'], (d.get('unprofiled_rerun') or {}).get('ms_per_step'), d['roofline']['kernel_ms_per_step'])" >> $out
  done
done
if [ -n "$1" ]; then timeout 500 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "$1" 2>&1 | tail -5 >> $out; fi
cat $out
