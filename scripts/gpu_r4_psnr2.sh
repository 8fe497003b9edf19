#!/bin/bash
# gpurun --timeout 1800 -- "bash scripts/gpu_r4_psnr2.sh"
# Second half of the 8x256 PSNR@iters study: the f16x3_train arm again for seeds 1-4 (its first run predates the range fix of
# k_wgrad_f16x3: NaN gradients at iteration 1257 of seed 2) and all three arms for seeds 5-8.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/psnr_8x256_b; mkdir -p $O
for s in 1 2 3 4; do
  timeout 300 python scripts/psnr_arms.py $s 2000 $O/f16_seed$s.json --arms engine_f16tr --hidden 256 --layers 8 --lr 1e-3 > $O/f16_seed$s.log 2>&1; echo "seed $s f16tr rc=$? $(tail -1 $O/f16_seed$s.log | cut -c1-160)"
done
for s in 5 6 7 8; do
  timeout 700 python scripts/psnr_arms.py $s 2000 $O/seed$s.json --arms engine,engine_f16tr,ref --hidden 256 --layers 8 --lr 1e-3 > $O/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed$s.log | cut -c1-160)"
done
