#!/bin/bash
# gpurun -- "bash scripts/gpu_bf16_wgrad_pmc.sh": SQ counters of the split-bf16 training kernels (bench.py --precision bf16x3_train)
cd /tmp && export TMPDIR=/tmp && mkdir -p $GRAFT_REPO_ROOT/gpurun_out && R=/tmp/pmcw && rm -rf $R && mkdir -p $R
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision bf16x3_train"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/sq -- $B > $R/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $R/sq2 -- $B > $R/sq2.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/bf16x3_train_pmc.txt 2>&1
import csv, glob, collections
for d in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("/tmp/pmcw/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "bf16x3" not in k or "pack" in k or "reduce" in k: continue
            k = k[k.index("k_"):].split("(")[0][:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print("    %-28s %16.0f per launch (%d launches)" % (c, x / cnt[(k, c)], cnt[(k, c)]))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/bf16x3_train_pmc.txt; tail -2 $R/sq.log | cut -c1-200
