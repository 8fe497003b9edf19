#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py --steps 10 --warmup 3 > $R/bench_dma.log 2>&1
NERFHIP_STAGE=reg timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_reg.log 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.csv" -size +30M -delete
grep -E "passed|failed" $R/pytest_gpu.log | tail -3; tail -2 $R/smoke.log; tail -1 $R/bench_dma.log | cut -c1-1500; tail -1 $R/bench_reg.log | cut -c1-900; ls -la $R/pmc_sq $R/pmc_fetch $R/pmc_write 2>&1 | head -20; tail -3 $R/pmc_sq.log
