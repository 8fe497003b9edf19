#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_r2_tests.sh [pytest args]"
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1400 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40
