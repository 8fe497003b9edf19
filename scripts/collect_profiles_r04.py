"""Copies the judged summaries of scripts/gpu_r4_final.sh (+ the PSNR and probe runs) from gpurun_out/ (scratch) into profiles/
(tracked):  python scripts/collect_profiles_r04.py"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P, tag = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"), "r04"


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def maybe(fn):
    try:
        fn()
    except (OSError, IndexError, KeyError, ValueError) as e:
        print("skipped:", repr(e)[:160])


def bench_lines():
    for src, dst in (("bench", "bench_line"), ("bench_f16x3_train", "bench_line_f16x3_train"), ("bench_f16x3_fwd_dgrad", "bench_line_f16x3_fwd_dgrad"),
                     ("bench_f16x3_fwd", "bench_line_f16x3_fwd"), ("bench_bf16x3_train", "bench_line_bf16x3_train"),
                     ("bench_fp32+bf16x3_train", "bench_line_fp32+bf16x3_train"), ("bench_fp32+f16x3_train", "bench_line_fp32+f16x3_train"),
                     ("bench_fern", "bench_line_fern_4x64"), ("bench_4x128", "bench_line_4x128"), ("bench_f16x3_train_4x128", "bench_line_f16x3_train_4x128"),
                     ("bench_eval", "bench_line_eval_800x800"), ("bench_eval_f16x3", "bench_line_eval_800x800_f16x3"),
                     ("bench_eval_bf16x3", "bench_line_eval_800x800_bf16x3")):
        p = os.path.join(G, src + ".log")
        if os.path.exists(p):
            try:
                json.dump(last_json_line(p), open(os.path.join(P, "%s_%s.json" % (tag, dst)), "w"), indent=1)
            except (IndexError, ValueError) as e:
                print("no JSON line in", src, repr(e)[:80])


def kernel_stats(src_dir, dst, cmd):
    path = os.path.join(G, src_dir, "bench_kernel_stats.csv")
    rows = list(csv.DictReader(open(path)))
    per = {}
    for r in csv.DictReader(open(os.path.join(G, src_dir, "bench_kernel_trace.csv"))):
        per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    timed = {}
    for k, v in per.items():
        v.sort()
        keep = v[len(v) * 3 // 23:] if len(v) >= 23 else v   # without the 3 warm-up steps of 23
        timed[k] = sum(d for _, d in keep) / len(keep)
    with open(os.path.join(P, "%s_%s" % (tag, dst)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline%s   (20 steps after 3 warm-up: the same run length as the bench line)\n" % cmd)
        f.write("%-72s %8s %14s %12s %8s %16s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "avg_ns_20_timed"))
        for r in rows:
            f.write("%-72s %8s %14s %12.0f %8s %16.0f\n" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"],
                                                           timed.get(r["Name"], 0.0)))


def parity():
    full = {}
    for f in sorted(glob.glob(os.path.join(G, "parity_fullsize_*.json"))):
        full[os.path.basename(f)[len("parity_fullsize_"):-5]] = json.load(open(f))
    json.dump(full, open(os.path.join(P, tag + "_parity_fullsize.json"), "w"), indent=1, sort_keys=True)
    shutil.copyfile(os.path.join(G, "parity_small_cases.json"), os.path.join(P, tag + "_parity_small_cases.json"))


def gpu_tests():
    with open(os.path.join(P, tag + "_gpu_tests.txt"), "w") as f:
        f.write("# python -m pytest tests -m gpu -q ; python __graft_entry__.py smoke   (MI355X; the round's last build)\n")
        f.write("".join(open(os.path.join(G, "pytest_gpu.log")).readlines()[-5:]))
        f.write("".join(open(os.path.join(G, "smoke.log")).readlines()[-2:]))


maybe(bench_lines)
maybe(lambda: kernel_stats("prof", "bench_kernel_stats.txt", ""))
maybe(lambda: kernel_stats("prof_f16", "bench_kernel_stats_f16x3_train.txt", " --precision f16x3_train"))
maybe(lambda: kernel_stats("prof_fern", "bench_kernel_stats_fern_4x64.txt", " --workload fern"))
maybe(parity)
maybe(gpu_tests)
for src, dst in (("pmc_summary_8x256_4096.txt", "pmc_summary.txt"), ("pmc_summary_8x256_4096.json", "pmc_summary_8x256_4096.json"),
                 ("pmc_summary_f16x3_train.txt", "pmc_summary_f16x3_train.txt"), ("pmc_summary_f16x3_train.json", "pmc_summary_f16x3_train.json"),
                 ("pmc_summary_fern.txt", "pmc_summary_fern_4x64.txt"), ("pmc_summary_fern.json", "pmc_summary_fern_4x64.json"),
                 ("f16_probe.txt", "f16_mfma_probe.txt"), ("r4_ab.txt", "dense_stash_ab.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, "%s_%s" % (tag, dst)))
print("\n".join(sorted(f for f in os.listdir(P) if f.startswith(tag))))
