#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_r4_x3_tests.sh"   -- every split-precision GPU test, no -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "x3" > $R/pytest_x3.log 2>&1; echo "pytest rc=$?" >> $R/pytest_x3.log
grep -E "passed|failed|rc=" $R/pytest_x3.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_x3.log | head -20; grep -E "^E   " $R/pytest_x3.log | cut -c1-400 | head -30
