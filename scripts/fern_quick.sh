#!/bin/bash
# gpurun -- "bash scripts/fern_quick.sh [pmc]": the fern lines of the fused backward, one JSON summary per line (and, with `pmc`, its counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "fused_backward" 2>&1 | tail -2
for a in "--compact fused --overlap 0" "--compact fused" "--compact fused_compact" ""; do
  python bench.py --workload fern --no-cpu-baseline --no-labelled-lines $a 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$a |', d['ms_per_step'], 'unprofiled', d['unprofiled_rerun']['ms_per_step'], 'issue', d['host_issue_ms_per_step'], d['unprofiled_rerun']['host_issue_ms_per_step'], {k:(v['ms_per_step'], v['frac']) for k,v in r['mlp_kernels'].items()}); print('    ', r['kernel_ms_per_step'])"
done
if [ "$1" = "pmc" ]; then bash scripts/gpu_r6.sh fused_pmc 2>&1 | grep -v "^tail"; fi
