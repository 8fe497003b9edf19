#!/bin/bash
# gpurun --timeout T -- "bash scripts/gpu_r6.sh PART [PART ...]"      (round 6; every sub-command under its own `timeout`)
#   tests    the GPU suite (pytest -m gpu) and smoke() on the library in the tree                 -> gpurun_out/r06/pytest_gpu.log, smoke.log
#   bench    the driver's default line incl. its labelled child lines (compacted, fern, eval, 4x128, trained regime) -> r06/bench.log
#   trained  scripts/bench_trained.py: 2000 iterations on the teacher scene, then {fp32, f16x3_train} x {dense, compacted, recomputed}
#   prof     rocprofv3 --kernel-trace --stats: the default line; the trained-regime arms dense / compacted / recomputed (weights of `trained`)
#   pmc      PMC passes (SQ / FETCH_SIZE / WRITE_SIZE, each its own run) of the default line and of the trained-regime arms
#   ab       dense vs compacted on the default workload (0 % zero rows: what the gather costs)
#   trained_more  bench_trained.py with the two-stream step, and on the 4x128 nets
#   fused    the fused backward of 64-wide nets (csrc/mlp64r.hip): its GPU tests, fern lines dense / fused / fused over the list, rocprofv3 stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT/gpurun_out/r06
W=$R/trained_weights.pt
line() {  # value, ms/step, kernels of the last JSON line of a log
  python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("zero_cotangent_fraction", {}).get("backward_sample_points"),
          {k: (v["ms_per_step"], v["frac"], v["hbm_frac"]) for k, v in (d["roofline"] or {}).get("mlp_kernels", {}).items()})
    for k, v in d.get("labelled_lines", {}).items():
        if k == "trained_regime":
            print("   trained_regime", {a: (b.get("value"), b.get("ms_per_step"), b.get("zero_cotangent_fraction")) for a, b in v.items() if isinstance(b, dict) and "value" in b}, v.get("error"))
        else:
            print("   %-22s" % k, v.get("value"), v.get("ms_per_step"), v.get("zero_cotangent_fraction"), v.get("dominant_kernel"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "unparsed", repr(e)[:200])
PY
}
need_weights() {
  [ -f $W ] || timeout 400 python scripts/bench_trained.py $R/bench_trained.json --iters 2000 --save-weights $W > $R/bench_trained.log 2>&1
}
P=/tmp/pmc6
run_pmc() {  # name, command...
  n=$1; shift
  mkdir -p $P/$n
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/$n/sq -- "$@" > $P/$n/sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$n/fetch -- "$@" > $P/$n/fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$n/write -- "$@" > $P/$n/write.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $P/$n $R/pmc_summary_$n.json > $R/pmc_summary_$n.txt 2>&1
  echo "== pmc $n"; cat $R/pmc_summary_$n.txt | cut -c1-220
}
for part in "$@"; do
case $part in
tests)
  timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
  timeout 200 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
  cp gpurun_out/parity_small_cases.json $R/ 2>/dev/null; for f in gpurun_out/parity_fullsize_*.json; do cp $f $R/ 2>/dev/null; done
  grep -E "passed|failed|rc=" $R/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $R/pytest_gpu.log | head -20; tail -2 $R/smoke.log ;;
bench)
  timeout 900 python bench.py > $R/bench.log 2>&1; echo "bench rc=$?"; line $R/bench.log ;;
trained)
  timeout 500 python scripts/bench_trained.py $R/bench_trained.json --iters 2000 --save-weights $W > $R/bench_trained.log 2>&1; echo "trained rc=$?"; tail -8 $R/bench_trained.log ;;
trained_more)   # the two-stream step in the compacted modes; the 4x128 nets train_nerf.py really builds
  timeout 400 python scripts/bench_trained.py $R/bench_trained_overlap1.json --iters 2000 --overlap 1 --arms fp32_compacted,f16x3_train_compacted,f16x3_train_recomputed,f16x3_train_dense > $R/bench_trained_overlap1.log 2>&1; tail -5 $R/bench_trained_overlap1.log
  timeout 400 python scripts/bench_trained.py $R/bench_trained_4x128.json --iters 2000 --hidden 128 --layers 4 > $R/bench_trained_4x128.log 2>&1; tail -9 $R/bench_trained_4x128.log
  # ... and 4 x 64 students (config/fern.yml's geometry on the teacher scene): dense / compacted / fused / fused over the list
  timeout 400 python scripts/bench_trained.py $R/bench_trained_4x64.json --iters 2000 --hidden 64 --layers 4 --arms fp32_dense,fp32_compacted,fp32_fused,fp32_fused_compact,fp32_fused_stash,fp32_auto,f16x3_train_dense,f16x3_train_recomputed > $R/bench_trained_4x64.log 2>&1; tail -9 $R/bench_trained_4x64.log ;;
fused)
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "fused or fern_declared_4x64_full" > $R/pytest_fused.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed" $R/pytest_fused.log | tail -2; grep -E "^FAILED|^ERROR|Error|assert" $R/pytest_fused.log | head -20
  for a in "" "--compact fused" "--compact dense" "--compact fused_compact" "--overlap 0" "--compact fused --overlap 0" "--compact dense --overlap 0"; do   # ("": fused over the stash, the default of these nets; "fused": the recomputing variant)
    t=$(echo $a | tr -d " -"); timeout 150 python bench.py --workload fern --no-cpu-baseline --no-labelled-lines $a > $R/fern_$t.log 2>&1; line $R/fern_$t.log
  done
  cd /tmp
  rm -rf $R/prof_fern_fused
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fern_fused -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload fern --overlap 0 --no-cpu-baseline --no-labelled-lines > $R/prof_fern_fused.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $R/prof_fern_fused -name "*kernel_stats.csv" | head -1); echo "== $f"; [ -n "$f" ] && head -14 $f | cut -c1-200 ;;
fused_pmc)   # counters of the fused kernel: matrix-pipe busy, wait buckets, LDS bank conflicts (each group its own pass)
  cd /tmp; P=/tmp/pmcf && rm -rf $P && mkdir -p $P
  CMD="python $GRAFT_REPO_ROOT/bench.py --workload fern --overlap 0 --steps 3 --warmup 1 --no-cpu-baseline --no-labelled-lines"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq -- $CMD > $P/sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $P/lds -- $CMD > $P/lds.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $CMD > $P/fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- $CMD > $P/write.log 2>&1
  cd $GRAFT_REPO_ROOT
  python - $P <<'PY' | tee $R/pmc_fused.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "bwd64r" if "k_bwd64r<" in r["Kernel_Name"] else ("fwd64r" if "k_fwd64r" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    a = {n: acc[k][n] / cnt[k][n] for n in acc[k]}
    print(k, {n: round(v, 1) for n, v in sorted(a.items())})
    wc = a.get("SQ_WAVE_CYCLES", 0); gui = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if wc and gui:
        print("   mfma_util %.3f wait_any %.3f wait_inst %.3f active %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4), a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc))
    if a.get("SQ_LDS_IDX_ACTIVE"):
        print("   lds bank conflict cycles / lds active cycles %.3f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]))
    if "FETCH_SIZE" in a:   # KiB; FETCH_SIZE counts 64 B per 128-B request for wide reads on gfx950: x2 (MI355X_MICROARCH.md)
        print("   HBM per launch: FETCH %.3f GB (x2 corr %.3f), WRITE %.3f GB" % (a["FETCH_SIZE"] * 1024 / 1e9, 2 * a["FETCH_SIZE"] * 1024 / 1e9, a["WRITE_SIZE"] * 1024 / 1e9))
PY
  ;;
ab)
  for a in "" "--compact" "--compact recompute" "--precision f16x3_train" "--precision f16x3_train --compact" "--precision f16x3_train --compact recompute"; do
    t=$(echo $a | tr -d " -"); timeout 150 python bench.py --no-cpu-baseline --no-labelled-lines $a > $R/ab_$t.log 2>&1; line $R/ab_$t.log
  done ;;
prof)
  need_weights
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-labelled-lines > $R/prof_default.log 2>&1
  for arm in f16x3_train_dense f16x3_train_compacted f16x3_train_recomputed fp32_compacted; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$arm -o trained -- python $GRAFT_REPO_ROOT/scripts/bench_trained.py $R/prof_$arm.json --load-weights $W --arms $arm > $R/prof_$arm.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  for d in prof_default prof_f16x3_train_dense prof_f16x3_train_compacted prof_f16x3_train_recomputed prof_fp32_compacted; do
    f=$(find $R/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; [ -n "$f" ] && head -12 $f | cut -c1-200
  done ;;
pmc)
  need_weights
  cd /tmp
  P=/tmp/pmc6 && rm -rf $P && mkdir -p $P
  run_pmc 8x256_4096 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-labelled-lines --overlap 0
  # BASELINE configs[3] (fern, 4 x 64 nets on the stashed fused backward): k_fwd64r writes the register-image stash, k_bwd64r reads it
  run_pmc 4x64_4096 python $GRAFT_REPO_ROOT/bench.py --workload fern --steps 3 --warmup 1 --no-cpu-baseline --no-labelled-lines --overlap 0
  run_pmc 4x128_4096 python $GRAFT_REPO_ROOT/bench.py --hidden 128 --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --no-labelled-lines --overlap 0
  for arm in f16x3_train_dense f16x3_train_compacted fp32_dense fp32_compacted; do
    run_pmc trained_$arm python $GRAFT_REPO_ROOT/scripts/bench_trained.py /tmp/pmc6/$arm.json --load-weights $W --arms $arm --steps 2 --warmup 1
  done
  cp $R/pmc_summary_8x256_4096.json $R/pmc_summary.json; cd $GRAFT_REPO_ROOT ;;
pmc_fern)   # only the fern passes of `pmc`
  cd /tmp; rm -rf $P/4x64_4096; mkdir -p $P
  run_pmc 4x64_4096 python $GRAFT_REPO_ROOT/bench.py --workload fern --steps 3 --warmup 1 --no-cpu-baseline --no-labelled-lines --overlap 0
  cd $GRAFT_REPO_ROOT ;;
pmc_4x128)  # the 4 x 128 nets train_nerf.py:117-134 builds (the labelled lines 4x128_fp32*)
  cd /tmp; rm -rf $P/4x128_4096; mkdir -p $P
  run_pmc 4x128_4096 python $GRAFT_REPO_ROOT/bench.py --hidden 128 --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --no-labelled-lines --overlap 0
  cd $GRAFT_REPO_ROOT ;;
*) echo "unknown part $part" ;;
esac
done
