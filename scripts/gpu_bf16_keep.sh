#!/bin/bash
# gpurun -- "bash scripts/gpu_bf16_keep.sh": parity of the split-bf16 kernels (twice: a race would be intermittent) + A/B against -DNHB_STASH_WAIT_ALL
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2; do timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bf16" 2>&1 | tail -1; done
bash scripts/gpu_ab.sh 'base waitall' --precision bf16x3_train --steps 10
