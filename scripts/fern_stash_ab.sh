#!/bin/bash
# gpurun -- "bash scripts/fern_stash_ab.sh": the fused backward of 64-wide nets, recomputing (mode 3) against stashed (mode 5), on the
# fern workload: the GPU parity test of every fused mode first, then both lines with and without the two-stream step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06s
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_backward_of_64" 2>&1 | tail -5 | tee gpurun_out/r06s/parity.txt
for a in "--compact fused --overlap 0" "--compact fused_stash --overlap 0" "--compact fused" "--compact fused_stash"; do
  tag=$(echo "$a" | tr -d ' -')
  timeout 300 python bench.py --workload fern --no-cpu-baseline --no-labelled-lines $a > gpurun_out/r06s/fern_$tag.log 2>&1
  tail -1 gpurun_out/r06s/fern_$tag.log > gpurun_out/r06s/fern_$tag.json
  python - "$a" gpurun_out/r06s/fern_$tag.json <<'PY' | tee -a gpurun_out/r06s/summary.txt
import sys, json
d = json.loads(open(sys.argv[2]).read()); r = d['roofline']
print(sys.argv[1], '|', d['ms_per_step'], 'unprofiled', d['unprofiled_rerun']['ms_per_step'],
      {k: (v['ms_per_step'], v['frac'], v.get('hbm_frac')) for k, v in r['mlp_kernels'].items()})
PY
done
