#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_psnr.sh"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 500 python scripts/psnr_curve.py 1500 4096 > gpurun_out/psnr.log 2>&1; echo "rc=$?" >> gpurun_out/psnr.log
tail -12 gpurun_out/psnr.log | cut -c1-400
