#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 400 python scripts/diag.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?" >> gpurun_out/diag.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_r1 -name "*.db" -size +20M -delete 2>/dev/null
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; tail -2 gpurun_out/smoke.log; cat gpurun_out/diag.log | cut -c1-400; tail -2 gpurun_out/bench_prof.log | cut -c1-300; ls -R gpurun_out/prof_r1 | head -20
