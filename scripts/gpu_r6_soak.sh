#!/bin/bash
# gpurun --timeout 3300 -- "bash scripts/gpu_r6_soak.sh"
# Round 6: (a) the GPU suite + smoke on the build, (b) the 20 000-iteration soak of VERDICT r5 item 2 on it (scripts/psnr_soak.py):
# two seeds x {f16x3_train, fp32} with the compacted backward (the f16x3_train arm in its stash-recomputing form, SOAK_F16_MODE), and the
# f16x3_train arm with the dense backward.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06_soak
R=$GRAFT_REPO_ROOT/gpurun_out/r06_soak
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $R/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $R/gpu_tests.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $R/smoke.txt
  cp gpurun_out/parity_small_cases.json $R/ 2>/dev/null; cp gpurun_out/parity_fullsize.json $R/ 2>/dev/null
fi
ITERS=${SOAK_ITERS:-20000}
timeout 300 python scripts/psnr_soak.py 9 40 $R/preflight.json --arms engine_f16tr,engine --check 40 --diag 20 --compact > $R/preflight.log 2>&1
rc=$?; echo "preflight rc=$rc"; tail -3 $R/preflight.log
if [ $rc -ne 0 ]; then tail -30 $R/preflight.log; exit 1; fi
for seed in 1 2; do
  timeout 400 python scripts/psnr_soak.py $seed $ITERS $R/soak_compact_seed$seed.json --arms engine --compact > $R/soak_compact_seed$seed.log 2>&1
  echo "compact fp32 soak seed $seed rc=$?"; grep "val_psnr" $R/soak_compact_seed$seed.log | tail -1
  timeout 300 python scripts/psnr_soak.py $seed $ITERS $R/soak_${SOAK_F16_MODE:-recompute}_seed$seed.json --arms engine_f16tr --compact ${SOAK_F16_MODE:-recompute} > $R/soak_${SOAK_F16_MODE:-recompute}_seed$seed.log 2>&1
  echo "${SOAK_F16_MODE:-recompute} f16x3 soak seed $seed rc=$?"; grep "val_psnr" $R/soak_${SOAK_F16_MODE:-recompute}_seed$seed.log | tail -1
  timeout 500 python scripts/psnr_soak.py $seed $ITERS $R/soak_dense_seed$seed.json --arms engine_f16tr > $R/soak_dense_seed$seed.log 2>&1
  echo "dense f16x3 soak seed $seed rc=$?"; grep "val_psnr" $R/soak_dense_seed$seed.log | tail -1
done
