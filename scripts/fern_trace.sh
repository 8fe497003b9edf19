#!/bin/bash
# gpurun -- "bash scripts/fern_trace.sh": rocprofv3 kernel trace of the fern line with the fused backward WITHOUT the per-launch events
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --workload fern --compact fused --overlap 0 --no-kernel-profile --no-cpu-baseline --no-labelled-lines --steps 20 --warmup 5 > /tmp/tr.log 2>&1
tail -1 /tmp/tr.log | cut -c1-300
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 5 steps: find k_select_rays launches as step markers
idx = [i for i, r in enumerate(rows) if "k_select_rays" in r["Kernel_Name"]]
a, b = idx[-6], idx[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("5 steps: wall %.3f ms/step, kernels busy %.3f ms/step, %d launches/step" % ((t1 - t0) / 5e6, busy / 5e6, len(seg) / 5))
prev = None
for r in seg[:len(seg) // 5]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-50s dur %7.1f us  gap before %6.1f us" % (r["Kernel_Name"][:50], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
PY
