#!/bin/bash
# Runs on the GPU box via:  gpurun --timeout 600 -- "bash scripts/gpu_ab_wgrad.sh"
# Weight-gradient kernel tuning inside ONE box (box-to-box variance is ~3 %): backward parity tests, per-workgroup
# timelines, bench with the built-in and with timeline-fitted split-K costs (8x256 and 4x128 nets).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider -k "edge or backward or northstar or e2e or select or image or python_api or linearity" > $R/pytest_ab.log 2>&1; echo "pytest rc=$?" >> $R/pytest_ab.log
timeout 120 python scripts/wgrad_timeline.py 786432 256 8 $R/costs_256.txt > $R/timeline_256.txt 2>&1
timeout 120 python scripts/wgrad_timeline.py 786432 128 4 $R/costs_128.txt > $R/timeline_128.txt 2>&1
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5"
timeout 120 $B > $R/ab_builtin.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_256.txt) timeout 120 $B > $R/ab_fit.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_256.txt) timeout 120 python scripts/wgrad_timeline.py 786432 256 8 > $R/timeline_256_fit.txt 2>&1
timeout 100 $B --hidden 128 --layers 4 > $R/ab_builtin_128.log 2>&1
NERFHIP_WGRAD_COSTS=$(cat $R/costs_128.txt) timeout 100 $B --hidden 128 --layers 4 > $R/ab_fit_128.log 2>&1
grep -E "passed|failed" $R/pytest_ab.log | tail -2
for f in ab_builtin ab_fit ab_builtin_128 ab_fit_128; do echo "$f: $(tail -1 $R/$f.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]; print(d["value"], d["ms_per_step"], {a:b for a,b in k.items() if b>0.2})
except Exception as e: print("ERR", e)')"; done
for f in timeline_256 timeline_256_fit timeline_128; do echo $f; grep -E "nwg|us per sample|fitted" $R/$f.txt | cut -c1-330; done
