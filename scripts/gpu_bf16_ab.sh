#!/bin/bash
# gpurun -- "bash scripts/gpu_bf16_ab.sh 'libnerfhip.so libnerfhip_late.so ...'": scripts/bf16x3_timing.py per library, two rounds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/bf16x3_ab.txt; : > $out
for round in 1 2; do for lib in $1; do timeout 120 python scripts/bf16x3_timing.py 786432 $lib 2>&1 | grep -E "^#|bf16x3 " >> $out; done; done
cat $out | cut -c1-200
