"""Copies the judged summaries of round 6's GPU calls (scripts/gpu_r6.sh -> gpurun_out/r06/, scripts/gpu_r6_soak.sh -> gpurun_out/r06_soak/)
from gpurun_out/ (scratch) into profiles/ (tracked) as r06_*:   python scripts/collect_profiles_r6.py
Every record carries (or is stamped with) the fingerprint of the kernel sources its library was built from (bench.lib_sources_sha16); a
record whose fingerprint is not the tree's gets a `_stale_build` note instead of silently standing next to the others."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

G, S, P = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "gpurun_out", "r06_soak"), os.path.join(ROOT, "profiles")
TAG = "r06"
SHA = bench.lib_sources_sha16()


def maybe(fn):
    try:
        fn()
    except (OSError, IndexError, KeyError, ValueError) as e:  # a part that was not run
        print("skipped:", fn.__name__, repr(e)[:160])


def last_json_line(path):
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])


def stamp(j):
    j["_builder_run"] = "measured by the builder through gpurun (scripts/gpu_r6.sh); the driver's own run is BENCH_r06.json"
    if j.get("lib_sources_sha16") != SHA:
        j["_stale_build"] = "kernel sources %s, the tree is %s" % (j.get("lib_sources_sha16"), SHA)
    return j


def bench_lines():
    json.dump(stamp(last_json_line(os.path.join(G, "bench.log"))), open(os.path.join(P, TAG + "_bench_line.json"), "w"), indent=1)
    for path in sorted(glob.glob(os.path.join(G, "ab_*.log"))):
        name = os.path.basename(path)[3:-4] or "dense"
        json.dump(stamp(last_json_line(path)), open(os.path.join(P, "%s_bench_line_ab_%s.json" % (TAG, name)), "w"), indent=1)


def trained():
    for name in ("bench_trained", "bench_trained_overlap1", "bench_trained_4x128", "bench_trained_4x64"):
        path = os.path.join(G, name + ".json")
        if not os.path.exists(path):
            continue
        j = json.load(open(path))
        j["lib_sources_sha16"] = j.get("lib_sources_sha16") or SHA  # (the script loads the library of the tree it runs in)
        j["_builder_run"] = "scripts/bench_trained.py through gpurun (scripts/gpu_r6.sh trained / trained_more)"
        json.dump(j, open(os.path.join(P, "%s_%s.json" % (TAG, name)), "w"), indent=1)


def fern_lines():
    """scripts/gpu_r6.sh fused: BASELINE configs[3] with the fused one-kernel backward (the default of these nets), dense, over the list."""
    for path in sorted(glob.glob(os.path.join(G, "fern_*.log"))):
        name = os.path.basename(path)[5:-4] or "default"
        json.dump(stamp(last_json_line(path)), open(os.path.join(P, "%s_bench_line_fern_%s.json" % (TAG, name)), "w"), indent=1)
    if os.path.exists(os.path.join(G, "pmc_fused.txt")):
        with open(os.path.join(P, TAG + "_pmc_fused.txt"), "w") as f:
            f.write("# rocprofv3 --pmc passes (each counter group its own run) of python bench.py --workload fern --overlap 0 --steps 3 --warmup 1: the fused\n"
                    "# backward k_bwd64r and the resident forward k_fwd64r of the 4 x 64 nets, averages per launch (kernel sources %s)\n" % SHA)
            f.write(open(os.path.join(G, "pmc_fused.txt")).read())


def kernel_stats():
    for d, cmd in (("prof_default", "python bench.py --no-cpu-baseline --no-labelled-lines"),
                   ("prof_f16x3_train_dense", "python scripts/bench_trained.py --load-weights W --arms f16x3_train_dense"),
                   ("prof_f16x3_train_compacted", "python scripts/bench_trained.py --load-weights W --arms f16x3_train_compacted"),
                   ("prof_f16x3_train_recomputed", "python scripts/bench_trained.py --load-weights W --arms f16x3_train_recomputed"),
                   ("prof_fp32_compacted", "python scripts/bench_trained.py --load-weights W --arms fp32_compacted"),
                   ("prof_fern_fused", "python bench.py --workload fern --overlap 0 --no-cpu-baseline --no-labelled-lines")):
        files = glob.glob(os.path.join(G, d, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0])))
        with open(os.path.join(P, "%s_kernel_stats_%s.txt" % (TAG, d[5:])), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- %s   (kernel sources %s; W: the weights after 2000 iterations on the teacher scene;\n"
                    "# every launch of the process: warm-up steps, the one gradient-comparison step and the timed steps)\n" % (cmd, SHA))
            f.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
            for r in rows[:24]:
                f.write("%-110s %8s %14s %12.0f %8s\n" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))


def parity_records():
    merged = {}
    for path in sorted(glob.glob(os.path.join(G, "parity_fullsize_*.json"))):
        merged[os.path.basename(path)[len("parity_fullsize_"):-5]] = json.load(open(path))
    if merged:
        json.dump(merged, open(os.path.join(P, TAG + "_parity_fullsize.json"), "w"), indent=1, sort_keys=True)
    if os.path.exists(os.path.join(G, "parity_small_cases.json")):
        shutil.copyfile(os.path.join(G, "parity_small_cases.json"), os.path.join(P, TAG + "_parity_small_cases.json"))


def gpu_tests():
    with open(os.path.join(P, TAG + "_gpu_tests.txt"), "w") as f:
        f.write("# python -m pytest tests -m gpu -q ; python __graft_entry__.py smoke   (MI355X; ONE run, library built from kernel sources %s)\n" % SHA)
        f.write("".join(ln for ln in open(os.path.join(G, "pytest_gpu.log")).readlines()[-6:] if ln.strip()))
        f.write("".join(open(os.path.join(G, "smoke.log")).readlines()[-2:]))


def pmc():
    for path in sorted(glob.glob(os.path.join(G, "pmc_summary_*.txt")) + glob.glob(os.path.join(G, "pmc_summary_*.json")) +
                       [os.path.join(G, "pmc_summary.json")]):
        if os.path.exists(path):
            shutil.copyfile(path, os.path.join(P, "%s_%s" % (TAG, os.path.basename(path))))


def soak(pattern="soak_*_seed*.json", out="_psnr_soak.txt"):
    runs = sorted(glob.glob(os.path.join(S, pattern)))
    if not runs:
        raise OSError("no soak runs")
    os.makedirs(os.path.join(P, TAG + "_psnr_soak_runs"), exist_ok=True)
    for r in runs:
        shutil.copyfile(r, os.path.join(P, TAG + "_psnr_soak_runs", os.path.basename(r)))
    docs = [json.load(open(r)) for r in runs]
    with open(os.path.join(P, TAG + out), "w") as f:
        W = f.write
        d0 = docs[0]
        W("# scripts/psnr_soak.py (scripts/gpu_r6_soak.sh): %s students, %s, %d rays/iter, lr %g x 0.1^(i/250000), %d iterations on the teacher scene\n"
          % (d0["student"], d0["image"], d0["rays_per_iter"], d0["lr0"], d0["iters"]))
        W("# arms: engine = TrainEngine on fp32 plans; engine_f16tr = the same engine on NERFHIP_PRECISION_F16X3_TRAIN plans; backward: compacted\n"
          "# (set_backward_compaction(True): data and weight gradient over the samples whose d(loss)/d(raw) row is non-zero), recomputed (its\n"
          "# stash-recomputing form, set_backward_compaction('recompute'): the same gradient bit for bit, another data flow), dense, or -- 64-wide\n"
          "# fp32 nets -- fused / fused_compact (csrc/mlp64r.hip: one persistent kernel per net, over every sample / over the list).\n"
          "# validation PSNR = -10 log10(coarse_mse + fine_mse) on 3 whole held-out 400x400 views (train_nerf.py:258-260, :339-347)\n")
        W("# kernel sources of the runs: %s   (the tree: %s%s)\n\n" % (sorted({d.get("lib_sources_sha16") for d in docs}), SHA,
                                                                       "" if all(d.get("lib_sources_sha16") == SHA for d in docs) else "  <-- STALE: re-run on the final build"))
        its = sorted(docs[0]["arms"][next(iter(docs[0]["arms"]))]["checkpoints"], key=int)
        cols = []
        for d in docs:
            for arm in d["arms"]:
                cols.append((d["seed"], d["backward"], arm, d["arms"][arm]))
        W("%-7s" % "iter" + "".join("  s%d %-13s %-6s" % (s, b, "fp32" if a == "engine" else "f16x3") for s, b, a, _ in cols) + "\n")
        for i in its:
            W("%-7s" % i + "".join("  %26.3f" % c["checkpoints"][i]["val_psnr"] for _, _, _, c in cols) + "\n")
        W("\n")
        for s, b, a, c in cols:
            dg = c["diagnostics"]
            ks = sorted(dg, key=int)
            finite = all(dg[k]["grad_finite"] and dg[k]["loss_finite"] and dg[k].get("kernel_grad_finite", True) for k in ks)
            zf = [dg[k].get("zero_cotangent_fraction") for k in ks if dg[k].get("zero_cotangent_fraction")]
            kg = [dg[k]["kernel_grad_vs_torch_worst_rel"] for k in ks if dg[k].get("kernel_grad_vs_torch_worst_rel") is not None]
            W("seed %d %-13s %-6s: every gradient / loss finite at %d diagnostics: %s; wall %6.1f s for %s iterations" %
              (s, b, "fp32" if a == "engine" else "f16x3", len(ks), finite, c["checkpoints"][its[-1]]["train_wall_s"], its[-1]))
            if zf:
                W("; zero-cotangent fraction coarse %.2f-%.2f, fine %.2f-%.2f" % (min(z["coarse"] for z in zf), max(z["coarse"] for z in zf),
                                                                                 min(z["fine"] for z in zf), max(z["fine"] for z in zf)))
            if kg:
                W("; fp16-piece gradient vs torch fp32 on ReLU-margin-filtered samples, of max|g|: %.1e-%.1e" % (min(kg), max(kg)))
            W("\n")
        # f16x3 - fp32 with the compacted backward, per seed, at the last checkpoint; and compacted - dense for the f16x3 arm
        W("\n")
        last = its[-1]
        by = {(s, b, a): c["checkpoints"][last]["val_psnr"] for s, b, a, c in cols}
        for s in sorted({s for s, _, _, _ in cols}):
            f16 = next(((b, v) for (ss, b, a), v in by.items() if ss == s and a == "engine_f16tr" and b != "dense"), None)
            if (s, "compacted", "engine") in by and f16:
                W("seed %d at %s: f16x3 (%s) - fp32 (compacted) %+.3f dB" % (s, last, f16[0], f16[1] - by[(s, "compacted", "engine")]))
            if (s, "dense", "engine_f16tr") in by and f16:
                W(";  f16x3 %s - f16x3 dense %+.3f dB" % (f16[0], f16[1] - by[(s, "dense", "engine_f16tr")]))
            W("\n")
        if d0["student"] == "8x256":
            W("(round 5, dense, same seeds / data stream / protocol, profiles/r05_psnr_soak.txt: fp32 26.95 / 26.83 dB, f16x3 26.98 / 26.72 dB at 20 000; 543 s / 260 s)\n")
        else:   # fused / fused over the list against dense, per seed, at the last checkpoint
            for s in sorted({s for s, _, _, _ in cols}):
                if (s, "dense", "engine") in by:
                    W("seed %d at %s: " % (s, last) + ";  ".join("%s - dense %+.3f dB" % (b, v - by[(s, "dense", "engine")])
                                                                 for (ss, b, a), v in sorted(by.items()) if ss == s and b != "dense") + "\n")


def soak64():
    soak("soak64_*_seed*.json", "_psnr_soak_4x64.txt")


for fn in (bench_lines, trained, fern_lines, kernel_stats, parity_records, gpu_tests, pmc, soak, soak64):
    maybe(fn)
print("\n".join(sorted(f for f in os.listdir(P) if f.startswith(TAG))))
