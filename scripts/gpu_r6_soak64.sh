#!/bin/bash
# gpurun --timeout 1500 -- "bash scripts/gpu_r6_soak64.sh"
# Round 6: the long run of the fused one-kernel backward of 64-wide nets (csrc/mlp64r.hip: another summation order of every weight
# gradient, another data flow): 4 x 64 students on the teacher scene, 2 seeds x 20 000 iterations x {dense, fused over the register-image
# stash (these nets' default), fused with the forward recomputed, fused over the list}
# (scripts/psnr_soak.py, fp32 engine arm; same data stream and protocol as scripts/gpu_r6_soak.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06_soak
R=$GRAFT_REPO_ROOT/gpurun_out/r06_soak
ITERS=${SOAK_ITERS:-20000}
for seed in 1 2; do
  for mode in ${SOAK64_MODES:-dense fused_stash fused fused_compact}; do
    if [ $mode = dense ]; then c=""; else c="--compact $mode"; fi
    timeout 300 python scripts/psnr_soak.py $seed $ITERS $R/soak64_${mode}_seed$seed.json --arms engine --hidden 64 --layers 4 $c > $R/soak64_${mode}_seed$seed.log 2>&1
    echo "4x64 $mode seed $seed rc=$?"; grep "val_psnr" $R/soak64_${mode}_seed$seed.log | tail -1
  done
done
