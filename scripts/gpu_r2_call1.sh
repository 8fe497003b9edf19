#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_r2_call1.sh"   (round 2, first call: loop mock + baseline bench lines)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r2c1 && O=gpurun_out/r2c1
timeout 240 ./scripts/loop_mock > $O/loop_mock.txt 2>&1; echo "rc=$?" >> $O/loop_mock.txt
for r in 4096 2048 1024; do
  timeout 200 python bench.py --steps 20 --warmup 3 --rays $r --no-cpu-baseline > $O/bench_rays$r.json 2> $O/bench_rays$r.err; echo "rays $r rc=$?"
done
timeout 200 python bench.py --steps 20 --warmup 3 --hidden 128 --layers 4 --no-cpu-baseline > $O/bench_4x128.json 2> $O/bench_4x128.err; echo "4x128 rc=$?"
cat $O/loop_mock.txt; tail -n 3 $O/bench_rays*.json $O/bench_4x128.json | cut -c1-1500
