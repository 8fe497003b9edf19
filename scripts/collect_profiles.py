"""Copies the judged summaries of a scripts/gpu_suite.sh (+ gpu_pmc.sh) run from gpurun_out/ (scratch) into profiles/
(tracked):  python scripts/collect_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f if l.startswith("{")]
    return json.loads(lines[-1])


def copy(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, "%s_%s" % (tag, dst)))


def kernel_stats(src_dir, dst):
    path = os.path.join(G, src_dir, "bench_kernel_stats.csv")
    if not os.path.exists(path):
        return
    rows = list(csv.DictReader(open(path)))
    # the same trace without the warm-up launches (first 3 of 23 steps): what bench.py's own event timing covers
    per = {}
    for r in csv.DictReader(open(os.path.join(G, src_dir, "bench_kernel_trace.csv"))):
        per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    timed = {}
    for k, v in per.items():
        v.sort()
        keep = v[len(v) * 3 // 23:] if len(v) >= 23 else v
        timed[k] = sum(d for _, d in keep) / len(keep)
    with open(os.path.join(P, "%s_%s" % (tag, dst)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline%s   (20 steps after 3 warm-up: the same run length as the bench line)\n" %
                (" --hidden 128 --layers 4" if "128" in src_dir else ""))
        f.write("%-64s %8s %14s %12s %8s %16s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "avg_ns_20_timed"))
        for r in rows:
            f.write("%-64s %8s %14s %12.0f %8s %16.0f\n" % (r["Name"][:64], r["Calls"], r["TotalDurationNs"],
                                                           float(r["AverageNs"]), r["Percentage"], timed.get(r["Name"], 0.0)))


def maybe(fn):
    try:
        fn()
    except (OSError, IndexError, KeyError, ValueError) as e:  # a pass that was not run this round
        print("skipped:", repr(e)[:120])


def bench_lines():
    for src, dst in (("bench.log", "bench_line.json"), ("bench_4x128.log", "bench_line_4x128.json"),
                     ("bench_4x128_single.log", "bench_line_4x128_single_stream.json"), ("bench_4x64.log", "bench_line_4x64.json"),
                     ("bench_8x512.log", "bench_line_8x512.json"), ("bench_eval.log", "bench_line_eval_800x800.json"),
                     ("bench_eval_bf16x3.log", "bench_line_eval_800x800_bf16x3.json"),
                     ("bench_eval_fp32_4x128.log", "bench_line_eval_800x800_4x128.json"),
                     ("bench_eval_bf16x3_4x128.log", "bench_line_eval_800x800_4x128_bf16x3.json"),
                     ("bench_bf16x3_fwd.log", "bench_line_bf16x3_fwd.json"), ("bench_bf16x3_fwd_4x128.log", "bench_line_bf16x3_fwd_4x128.json"),
                     ("bench_bf16x3_fwd_dgrad.log", "bench_line_bf16x3_fwd_dgrad.json"),
                     ("bench_bf16x3_fwd_dgrad_4x128.log", "bench_line_bf16x3_fwd_dgrad_4x128.json"),
                     ("bench_bf16x3_train.log", "bench_line_bf16x3_train.json")):
        if os.path.exists(os.path.join(G, src)):
            json.dump(last_json_line(os.path.join(G, src)), open(os.path.join(P, "%s_%s" % (tag, dst)), "w"), indent=1)
    with open(os.path.join(P, tag + "_multi_rank_one_gpu.txt"), "w") as f:
        f.write("# The N > 1 paths of bench.py on the ONE GPU of the builder's box (both ranks on cuda:0, gloo: RCCL refuses two ranks\n"
                "# on one device): self-launch (`python bench.py --gpus 2`, no torch.distributed.run around it), weak scaling, strong\n"
                "# scaling (--image 800 --global-rays 8192 = BASELINE configs[2]) and the ray-sharded eval (--mode eval).  NOT a\n"
                "# measurement of scaling: two ranks share one GPU, so every per-rank time doubles.\n")
        for name in ("dp2_weak", "dp2_strong", "dp2_eval"):
            path = os.path.join(G, name + ".log")
            if os.path.exists(path):
                j = last_json_line(path)
                f.write("%-11s n_gpus %d  scaling %-6s  %9.0f rays/s  ms/step per rank %s  backend %s\n    %s\n" % (
                    name, j["n_gpus"], j["scaling"], j["value"], j["ms_per_step_per_rank"], j["config"].get("backend"), j["config"]["workload"]))


maybe(bench_lines)


def rays_and_overlap():
    with open(os.path.join(P, tag + "_bench_rays.txt"), "w") as f:
        f.write("# python bench.py --rays R --no-cpu-baseline  (per-GPU batch of a 2- / 4-GPU strong-scaling split of 4096 rays)\n")
        for r in (4096, 2048, 1024):
            j = last_json_line(os.path.join(G, "bench.log" if r == 4096 else "bench_rays%d.log" % r))
            f.write("rays/GPU %5d  %9.0f rays/s  %7.3f ms/step  step frac %.4f  kernels(ms/step) %s\n" %
                    (r, j["value"], j["ms_per_step"], j["step_frac_of_fp32_mfma_peak"],
                     {k: v for k, v in list(j["roofline"]["kernel_ms_per_step"].items())[:4]}))
    with open(os.path.join(P, tag + "_overlap_ab.txt"), "w") as f:
        f.write("# bench.py --overlap 0|1 (single-stream step | coarse backward on a second stream next to the fine pass), "
                "interleaved twice; kernel times overlap in the two-stream runs\n")
        for l in open(os.path.join(G, "overlap_ab.jsonl")):
            j = json.loads(l)
            f.write("%s two-stream %d  %9.0f rays/s  %7.3f ms/step  step frac %.4f\n" %
                    ("8x256" if "8x256" in j["config"]["workload"] else "4x128", int(j["config"].get("two_stream_step", -1)), j["value"],
                     j["ms_per_step"], j["step_frac_of_fp32_mfma_peak"]))


maybe(rays_and_overlap)
maybe(lambda: kernel_stats("prof", "bench_kernel_stats.txt"))
maybe(lambda: kernel_stats("prof128", "bench_kernel_stats_4x128.txt"))


def eval_kernel_stats():
    rows = list(csv.DictReader(open(os.path.join(G, "prof_eval_bf16x3", "bench_kernel_stats.csv"))))
    with open(os.path.join(P, tag + "_bench_kernel_stats_eval_bf16x3.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --mode eval --precision bf16x3 --no-cpu-baseline   (1 warm-up + 3 timed poses)\n")
        f.write("%-64s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows:
            f.write("%-64s %8s %14s %12.0f %8s\n" % (r["Name"][:64], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))


maybe(eval_kernel_stats)
for src, dst in (("eval.log", "eval_800x800.txt"), ("phase_timing.txt", "phase_timing.txt"), ("wgrad_timeline.txt", "wgrad_timeline.txt"),
                 ("wgrad_timeline_4x128.txt", "wgrad_timeline_4x128.txt"), ("pmc_summary_8x256_4096.txt", "pmc_summary.txt"),
                 ("pmc_summary_8x256_4096.json", "pmc_summary_8x256_4096.json"), ("pmc_summary_4x128_4096.txt", "pmc_summary_4x128_4096.txt"),
                 ("pmc_summary_4x128_4096.json", "pmc_summary_4x128_4096.json"), ("bf16x3_timing.txt", "bf16x3_timing.txt"),
                 ("bf16x3_pmc.txt", "bf16x3_pmc.txt")):
    copy(src, dst)


def gpu_tests():
    with open(os.path.join(P, tag + "_gpu_tests.txt"), "w") as f:
        f.write("# python -m pytest tests -m gpu -q ; python __graft_entry__.py smoke   (MI355X)\n")
        f.write("".join(open(os.path.join(G, "pytest_gpu.log")).readlines()[-4:]))
        f.write("".join(open(os.path.join(G, "smoke.log")).readlines()[-2:]))


maybe(gpu_tests)
print("\n".join(sorted(f for f in os.listdir(P) if f.startswith(tag))))
