#!/bin/bash
# gpurun --timeout 2300 -- "bash scripts/gpu_r5_soak.sh"
# Round 5, call 1: (a) the one-GPU proxy lines of VERDICT r4 item 7 (per-GPU shape of BASELINE configs[2]: 800x800, 1024 rays, and the
# whole 8192-ray step on one GPU), (b) the 20 000-iteration soak of item 3 (scripts/psnr_soak.py), two seeds x {f16x3_train, fp32}.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/soak
R=$GRAFT_REPO_ROOT/gpurun_out
for spec in "1024 fp32" "8192 fp32" "1024 f16x3_train" "8192 f16x3_train"; do
  set -- $spec
  timeout 120 python bench.py --no-cpu-baseline --image 800 --rays $1 --precision $2 > $R/bench_800_$1_$2.log 2>&1
  tail -1 $R/bench_800_$1_$2.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])
except Exception as e:
    print('$1 $2 unparsed', repr(e)[:120])
"
done
ITERS=${SOAK_ITERS:-20000}
# preflight: 40 iterations of both arms with a diagnostic and a validation pass (a script bug must not cost half an hour)
timeout 300 python scripts/psnr_soak.py 9 40 $R/soak/preflight.json --arms engine_f16tr,engine --check 40 --diag 20 > $R/soak/preflight.log 2>&1
rc=$?; echo "preflight rc=$rc"; tail -4 $R/soak/preflight.log
if [ $rc -ne 0 ]; then tail -30 $R/soak/preflight.log; exit 1; fi
for seed in 1 2; do
  timeout 1150 python scripts/psnr_soak.py $seed $ITERS $R/soak/soak_seed$seed.json --arms engine_f16tr,engine > $R/soak/soak_seed$seed.log 2>&1
  echo "soak seed $seed rc=$?"; grep -c diag $R/soak/soak_seed$seed.log; grep "val_psnr" $R/soak/soak_seed$seed.log | tail -4
done
