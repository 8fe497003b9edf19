#!/bin/bash
# Runs on the GPU box (via gpurun): gpu test-suite, smoke(), a short bench.  Everything under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; free -g | head -2) > gpurun_out/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 180 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench1.log
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench1.log
