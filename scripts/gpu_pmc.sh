#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_pmc.sh"
# PMC counters of the three MLP kernels in SEPARATE passes (SQ+GRBM / FETCH_SIZE / WRITE_SIZE), kernel trace only.
cd /tmp && export TMPDIR=/tmp && mkdir -p $GRAFT_REPO_ROOT/gpurun_out && R=/tmp/pmc && rm -rf $R && mkdir -p $R
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap 0 $PMC_BENCH_ARGS"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/sq -- $B > $R/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/fetch -- $B > $R/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/write -- $B > $R/write.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $R $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.json > $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt; tail -2 $R/sq.log | cut -c1-300
