#!/bin/bash
# gpurun --timeout 600 -- "bash scripts/gpu_pmc_vmem.sh"   -- VMEM-path counters of the forward kernel, training vs inference variant
cd /tmp && export TMPDIR=/tmp && R=/tmp/pv && rm -rf $R && mkdir -p $R $GRAFT_REPO_ROOT/gpurun_out
B="python $GRAFT_REPO_ROOT/scripts/fwd_train_vs_infer.py"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/a -- $B > $R/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_BUSY TCP_TCP_TA_DATA_STALL_CYCLES --output-format csv -d $R/b -- $B > $R/b.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pmc_vmem.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_mlp_fwd16' not in n: continue
        k = n[n.index('k_mlp_fwd16'):].split('(')[0]
        acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(acc):
    print(key)
    for c, v in sorted(acc[key].items()):
        print('   %-36s mean %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_vmem.txt | head -70; tail -3 $R/a.log | cut -c1-200
