#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench5.log 2>&1
timeout 200 python scripts/phase_timing.py > $R/phase.log 2>&1
grep -E "passed|failed" $R/pytest_gpu.log | tail -3; tail -2 $R/smoke.log; tail -1 $R/bench5.log | cut -c1-1400; grep -v amdgpu.ids $R/phase.log | tail -14
