#!/bin/bash
# gpurun --timeout 500 -- "bash scripts/gpu_r4_guests.sh"
# The guest blocks of the split-precision weight-gradient kernel on MI355X: the split-precision part of the GPU suite, then the A/B
# against the build without them (libnerfhip_noguest.so = plan.cpp -DNHW_NO_GUESTS).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "x3" > $R/pytest_guests.log 2>&1; echo "pytest rc=$?" >> $R/pytest_guests.log
grep -E "passed|failed|rc=" $R/pytest_guests.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_guests.log | head
bash scripts/gpu_r4_w2_diag.sh 'product noguest product noguest' '--precision f16x3_train'
cp gpurun_out/r4_w2_diag.txt gpurun_out/r4_guests_8x256.txt
bash scripts/gpu_r4_w2_diag.sh 'product noguest' '--hidden 128 --layers 4 --precision f16x3_train'
cp gpurun_out/r4_w2_diag.txt gpurun_out/r4_guests_4x128.txt
bash scripts/gpu_r4_w2_diag.sh 'product noguest' '--precision bf16x3_train'
