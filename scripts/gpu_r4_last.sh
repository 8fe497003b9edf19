#!/bin/bash
# gpurun --timeout 330 -- "bash scripts/gpu_r4_last.sh"
# The round's last GPU minutes: the default bench line once more (its PMC summary now carries this library's fingerprint), then the
# f16x3_train PSNR arm of scripts/psnr_arms.py on the two-wave kernels for as many seeds as fit (pairs with the fp32 engine arm of
# profiles/r04_psnr_8x256_runs/: same seeds, same draws).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/psnr_w2
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 python bench.py > $R/bench.log 2>&1; tail -1 $R/bench.log | cut -c1-300
for s in 1 2 3; do
  timeout 95 python scripts/psnr_arms.py $s 2000 $R/psnr_w2/seed$s.json --arms engine_f16tr --hidden 256 --layers 8 --lr 1e-3 > $R/psnr_w2/seed$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $R/psnr_w2/seed$s.log | cut -c1-200)"
done
