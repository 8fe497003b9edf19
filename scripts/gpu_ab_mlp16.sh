#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_ab_mlp16.sh"
# A/B inside one box: 32x32x2 kernels (mlp.hip, NERFHIP_MLP=32) vs 16x16x4 two-waves-per-SIMD kernels (mlp16.hip, default).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
NERFHIP_MLP=16 timeout 400 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_mlp16.log 2>&1; echo "pytest rc=$?" >> $R/pytest_mlp16.log
NERFHIP_MLP=32 timeout 400 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_mlp32.log 2>&1; echo "pytest rc=$?" >> $R/pytest_mlp32.log
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5"
NERFHIP_MLP=32 timeout 120 $B > $R/ab32.log 2>&1
NERFHIP_MLP=16 timeout 120 $B > $R/ab16.log 2>&1
NERFHIP_MLP=32 timeout 100 $B --hidden 128 --layers 4 > $R/ab32_128.log 2>&1
NERFHIP_MLP=16 timeout 100 $B --hidden 128 --layers 4 > $R/ab16_128.log 2>&1
NERFHIP_MLP=16 timeout 200 python scripts/eval_bench.py > $R/eval16.log 2>&1
grep -E "passed|failed" $R/pytest_mlp16.log $R/pytest_mlp32.log | tail -4
for f in ab32 ab16 ab32_128 ab16_128; do echo "$f: $(tail -1 $R/$f.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]; print(d["value"], d["ms_per_step"], {a:b for a,b in k.items() if b>0.2})
except Exception as e: print("ERR", e)')"; done
tail -1 $R/eval16.log | cut -c1-400
