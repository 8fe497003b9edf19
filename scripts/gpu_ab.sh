#!/bin/bash
# gpurun --timeout 900 -- "bash scripts/gpu_ab.sh 'VARIANT1 VARIANT2 ...' [bench args]"   ("base" = the product library); two interleaved rounds
cd "$GRAFT_REPO_ROOT"; V=$1; shift; O=gpurun_out/ab; mkdir -p $O; rm -f $O/ab.jsonl; L=""
for round in 1 2; do for v in $V; do
  n=$v; [ "$v" = "base" ] && n=""
  timeout 200 python scripts/ab_bench.py "$n" --steps 20 --warmup 3 --no-cpu-baseline --overlap 0 "$@" 2>> $O/ab.err >> $O/ab.jsonl; L="$L $v"
done; done
python scripts/bench_summary.py $L < $O/ab.jsonl
