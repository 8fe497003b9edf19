#!/bin/bash
# gpurun --timeout 300 -- "bash scripts/gpu_phase16.sh"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
NERFHIP_MLP=16 timeout 120 python scripts/phase_timing.py > gpurun_out/phase16.txt 2>&1
timeout 120 python scripts/phase_timing.py > gpurun_out/phase32.txt 2>&1
cat gpurun_out/phase16.txt gpurun_out/phase32.txt | grep -v amdgpu.ids
