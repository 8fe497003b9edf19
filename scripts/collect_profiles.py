"""Copies the judged summaries of the round's GPU calls (scripts/gpu_r5.sh, scripts/gpu_r5_soak.sh) from gpurun_out/ (scratch) into
profiles/ (tracked), every file named for the round:   python scripts/collect_profiles.py r05
Each bench line carries `lib_sources_sha16` (the kernel sources its library was built from); a line whose fingerprint is not the
tree's is written with a `_stale_build` note instead of silently standing next to the others (three builds under one tag)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def here_sha():
    import bench
    return bench.lib_sources_sha16()


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def maybe(fn):
    try:
        fn()
    except (OSError, IndexError, KeyError, ValueError) as e:  # a part that was not run
        print("skipped:", fn.__name__, repr(e)[:120])


def bench_lines():
    sha = here_sha()
    for src, dst in (("bench.log", "bench_line.json"), ("bench_f16x3_train.log", "bench_line_f16x3_train.json"),
                     ("bench_4x128.log", "bench_line_4x128.json"), ("bench_f16x3_train_4x128.log", "bench_line_f16x3_train_4x128.json"),
                     ("bench_fern_4x64.log", "bench_line_fern_4x64.json"), ("bench_fern_4x64_f16x3_train.log", "bench_line_fern_4x64_f16x3_train.json"),
                     ("bench_fern_4x64_f16x3_fwd.log", "bench_line_fern_4x64_f16x3_fwd.json"),
                     ("bench_eval.log", "bench_line_eval_800x800.json"), ("bench_eval_f16x3.log", "bench_line_eval_800x800_f16x3.json"),
                     ("bench_800_1024_fp32.log", "bench_line_800x800_1024rays.json"), ("bench_800_8192_fp32.log", "bench_line_800x800_8192rays.json"),
                     ("bench_800_1024_f16x3_train.log", "bench_line_800x800_1024rays_f16x3_train.json"),
                     ("bench_800_8192_f16x3_train.log", "bench_line_800x800_8192rays_f16x3_train.json")):
        path = os.path.join(G, src)
        if not os.path.exists(path):
            continue
        try:
            j = last_json_line(path)
        except (IndexError, ValueError):
            print("no json line in", src)
            continue
        j["_builder_run"] = "measured by the builder through gpurun (scripts/gpu_r5*.sh); the driver's own run is BENCH_r05.json"
        if j.get("lib_sources_sha16") != sha:
            j["_stale_build"] = "kernel sources %s, the tree is %s" % (j.get("lib_sources_sha16"), sha)
        json.dump(j, open(os.path.join(P, "%s_%s" % (tag, dst)), "w"), indent=1)


def kernel_stats(src_dir, dst, args):
    path = os.path.join(G, src_dir, "bench_kernel_stats.csv")
    rows = list(csv.DictReader(open(path)))
    per = {}
    for r in csv.DictReader(open(os.path.join(G, src_dir, "bench_kernel_trace.csv"))):
        per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    timed = {}
    for k, v in per.items():  # the same trace without the warm-up launches (first 3 of 23 steps): what bench.py's own event timing covers
        v.sort()
        keep = v[len(v) * 3 // 23:] if len(v) >= 23 else v
        timed[k] = sum(d for _, d in keep) / len(keep)
    with open(os.path.join(P, "%s_%s" % (tag, dst)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline%s   (20 steps after 3 warm-up: the same run length as the bench line)\n" % args)
        f.write("%-72s %8s %14s %12s %8s %16s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "avg_ns_20_timed"))
        for r in rows:
            f.write("%-72s %8s %14s %12.0f %8s %16.0f\n" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"],
                                                           timed.get(r["Name"], 0.0)))


def parity_records():
    merged = {}
    for path in sorted(glob.glob(os.path.join(G, "parity_fullsize_*.json"))):
        merged[os.path.basename(path)[len("parity_fullsize_"):-5]] = json.load(open(path))
    if merged:
        json.dump(merged, open(os.path.join(P, tag + "_parity_fullsize.json"), "w"), indent=1, sort_keys=True)
    if os.path.exists(os.path.join(G, "parity_small_cases.json")):
        shutil.copyfile(os.path.join(G, "parity_small_cases.json"), os.path.join(P, tag + "_parity_small_cases.json"))


def gpu_tests():
    with open(os.path.join(P, tag + "_gpu_tests.txt"), "w") as f:
        f.write("# python -m pytest tests -m gpu -q ; python __graft_entry__.py smoke   (MI355X; ONE run, library built from kernel sources %s)\n" % here_sha())
        f.write("".join(ln for ln in open(os.path.join(G, "pytest_gpu.log")).readlines()[-6:] if ln.strip()))
        f.write("".join(open(os.path.join(G, "smoke.log")).readlines()[-2:]))


maybe(bench_lines)
maybe(lambda: kernel_stats("prof", "bench_kernel_stats.txt", ""))
maybe(lambda: kernel_stats("prof_f16", "bench_kernel_stats_f16x3_train.txt", " --precision f16x3_train"))
maybe(parity_records)
maybe(gpu_tests)
for src, dst in (("pmc_summary_8x256_4096.txt", "pmc_summary.txt"), ("pmc_summary_8x256_4096.json", "pmc_summary.json"),
                 ("pmc_summary_f16x3_train.txt", "pmc_summary_f16x3_train.txt"), ("pmc_summary_f16x3_train.json", "pmc_summary_f16x3_train.json"),
                 ("mocks.txt", "ws_loop_mock.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, "%s_%s" % (tag, dst)))
print("\n".join(sorted(f for f in os.listdir(P) if f.startswith(tag))))
