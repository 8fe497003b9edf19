#!/bin/bash
# Runs on the GPU box via:  gpurun --timeout 2400 -- "bash scripts/gpu_suite.sh"
# gpu test-suite, smoke(), bench (default, smaller per-GPU batches, the 4x128 nets the reference's scripts really build,
# single- vs two-stream step), rocprofv3 kernel trace, 800x800 inference, and the `make dbg` build's stamp diagnostics.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 500 -p no:cacheprovider > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log
timeout 300 python bench.py > $R/bench.log 2>&1
timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline > $R/bench_4x128.log 2>&1
for r in 2048 1024; do timeout 200 python bench.py --rays $r --no-cpu-baseline > $R/bench_rays$r.log 2>&1; done
rm -f $R/overlap_ab.jsonl
for o in 0 1 0 1; do
  timeout 200 python bench.py --no-cpu-baseline --overlap $o 2>/dev/null | tail -1 >> $R/overlap_ab.jsonl
  timeout 200 python bench.py --hidden 128 --layers 4 --no-cpu-baseline --overlap $o 2>/dev/null | tail -1 >> $R/overlap_ab.jsonl
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/bench_prof.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof128 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --hidden 128 --layers 4 > $R/bench_prof128.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/eval_bench.py > $R/eval.log 2>&1
if [ -f nerf-pytorch_amd/libnerfhip_dbg.so ]; then
  timeout 200 python scripts/wgrad_timeline.py > $R/wgrad_timeline.txt 2>&1
  timeout 200 python scripts/phase_timing.py > $R/phase_timing.txt 2>&1
fi
grep -E "passed|failed" $R/pytest_gpu.log | tail -2; tail -2 $R/smoke.log; tail -1 $R/bench.log | cut -c1-2600; tail -1 $R/bench_4x128.log | cut -c1-700; tail -1 $R/bench_rays2048.log | cut -c1-300; tail -1 $R/bench_rays1024.log | cut -c1-300; tail -1 $R/eval.log | cut -c1-500; ls $R/prof | head
