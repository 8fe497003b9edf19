#!/bin/bash
# gpurun --timeout 300 -- "bash scripts/gpu_mfma_rate.sh"
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 120 ./scripts/mfma_rate > gpurun_out/mfma_rate.txt 2>&1; echo "rc=$?" >> gpurun_out/mfma_rate.txt
cat gpurun_out/mfma_rate.txt
