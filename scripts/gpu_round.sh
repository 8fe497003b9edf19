#!/bin/bash
# Runs on the GPU box via:  gpurun --timeout 1500 -- "bash scripts/gpu_round.sh"
# gpu suite + smoke, weight-gradient kernel A/B (register operands vs LDS-staged, provisional vs fitted split-K costs),
# per-workgroup timelines, eval bench with the PNG output stage, then the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 420 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 120 python scripts/wgrad_timeline.py 786432 256 8 $R/costs_256.txt > $R/timeline_lds_256.txt 2>&1
NERFHIP_WGRAD=reg timeout 120 python scripts/wgrad_timeline.py 786432 256 8 > $R/timeline_reg_256.txt 2>&1
timeout 120 python scripts/wgrad_timeline.py 786432 128 4 $R/costs_128.txt > $R/timeline_lds_128.txt 2>&1
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5"
NERFHIP_WGRAD=reg timeout 120 $B > $R/ab_reg.log 2>&1
timeout 120 $B > $R/ab_lds.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_256.txt) timeout 120 $B > $R/ab_lds_fit.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_256.txt) timeout 120 python scripts/wgrad_timeline.py 786432 256 8 > $R/timeline_lds_256_fit.txt 2>&1
NERFHIP_WGRAD=reg timeout 100 $B --hidden 128 --layers 4 > $R/ab_reg_128.log 2>&1
timeout 100 $B --hidden 128 --layers 4 > $R/ab_lds_128.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_128.txt) timeout 100 $B --hidden 128 --layers 4 > $R/ab_lds_fit_128.log 2>&1
timeout 200 python scripts/eval_bench.py > $R/eval.log 2>&1
timeout 300 python bench.py > $R/bench.log 2>&1
grep -E "passed|failed" $R/pytest_gpu.log | tail -2; tail -1 $R/smoke.log
for f in ab_reg ab_lds ab_lds_fit ab_reg_128 ab_lds_128 ab_lds_fit_128; do echo "$f: $(tail -1 $R/$f.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
except Exception as e: print("ERR", e)')"; done
tail -1 $R/eval.log | cut -c1-600; tail -3 $R/timeline_lds_256.txt | cut -c1-400
