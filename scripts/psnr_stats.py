"""Paired statistics of scripts/psnr_arms.py runs:  python scripts/psnr_stats.py gpurun_out/psnr_arms > profiles/r03_psnr_400.txt
Per checkpoint and arm pair: mean paired difference of validation PSNR over the usable seeds, its 95 % confidence
interval (Student t), and an exact two-sided sign test.  A seed is dropped (and reported) when one of its nets COLLAPSED in
any arm: validation coarse+fine PSNR below 12 dB at the last checkpoint -- one of the two nets renders a constant (dead
ReLUs under lr 5e-3: round 2 met it with the coarse net of seed 0, the 8x256 students of round 3 with the fine net of
seeds 3 and 4), which the reference's own path does from the same initial weights."""
import glob
import json
import math
import os
import sys

T95 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201, 12: 2.179}


def sign_test(diffs):
    n = sum(1 for d in diffs if d != 0)
    k = sum(1 for d in diffs if d > 0)
    if n == 0:
        return 1.0
    tail = sum(math.comb(n, j) for j in range(0, min(k, n - k) + 1)) / 2.0 ** n
    return min(1.0, 2.0 * tail)


def main(root):
    runs = {}
    for f in sorted(glob.glob(os.path.join(root, "seed*.json"))):
        j = json.load(open(f))
        runs[j["seed"]] = j
    if not runs:
        print("no runs under", root)
        return
    any_run = next(iter(runs.values()))
    print("# scripts/psnr_arms.py: students %s, %s, %d rays/iter, 64+128, %d iterations, initial lr %g (every arm); %d seeds: %s" % (
        any_run["student"], any_run["image"], any_run["rays_per_iter"], any_run["iters"], any_run.get("lr0", 5e-3), len(runs), sorted(runs)))
    last = str(max(int(k) for k in any_run["arms"]["engine"]))
    collapsed = []
    for s, j in sorted(runs.items()):
        for arm, h in j["arms"].items():
            if h[last]["val_psnr"] < 12.0:
                which = "fine" if h[last]["val_psnr_fine"] < 12.0 else "coarse"
                collapsed.append((s, arm, which, round(h[last]["val_psnr"], 2), round(h[last]["val_psnr_fine"], 2),
                                  round(h[last].get("val_psnr_coarse", float("nan")), 2)))
    bad = sorted({c[0] for c in collapsed})
    print("# collapsed nets (validation PSNR < 12 dB at iteration %s) as (seed, arm, net, val, val fine-only, val coarse-only): %s"
          % (last, collapsed or "none"))
    print("# seeds dropped from the paired statistics: %s" % (bad or "none"))
    use = [s for s in sorted(runs) if s not in bad]
    per_arm = {}
    for s, arm, which, *_ in collapsed:
        per_arm.setdefault(arm, []).append((s, which))
    all_arms = sorted({a for j in runs.values() for a in j["arms"]})
    print("# collapse rate per arm: " + ", ".join("%s %d/%d" % (a, len(per_arm.get(a, [])), sum(1 for j in runs.values() if a in j["arms"])) for a in all_arms))
    arms = [a for a in ("ref", "dropin", "engine_td", "engine", "engine_bf16fwd", "engine_bf16fd", "engine_bf16tr", "engine_f16fd", "engine_f16tr")
            if any(a in runs[s]["arms"] for s in use)]
    checks = sorted(int(k) for k in any_run["arms"]["engine"])
    print("\n== validation PSNR (coarse+fine), mean over usable seeds [n] ==")
    print("%-10s" % "iteration" + "".join("%16s" % a for a in arms))
    for c in checks:
        row = "%-10d" % c
        for a in arms:
            v = [runs[s]["arms"][a][str(c)]["val_psnr"] for s in use if a in runs[s]["arms"]]
            row += "%11.3f [%2d]" % (sum(v) / len(v), len(v)) if v else "%16s" % "-"
        print(row)
    pairs = [("engine", "ref"), ("engine", "dropin"), ("engine", "engine_td"), ("engine_td", "dropin"), ("dropin", "ref")]
    extra = [(a, "engine") for a in ("engine_bf16fwd", "engine_bf16fd", "engine_bf16tr", "engine_f16fd", "engine_f16tr") if any(a in runs[s]["arms"] for s in use)]
    extra += [(a, "ref") for a in ("engine_f16fd", "engine_f16tr") if any(a in runs[s]["arms"] and "ref" in runs[s]["arms"] for s in use)]
    if extra:
        pairs = extra + [p for p in pairs if all(any(a in runs[s]["arms"] for s in use) for a in p)]
    for key, label in (("val_psnr", "validation PSNR coarse+fine"), ("val_psnr_fine", "validation PSNR, fine net alone"),
                       ("train_psnr", "training-batch PSNR (last 50 iterations)")):
        print("\n== paired differences, %s: mean [95 %% CI]  (n; sign test p) ==" % label)
        print("%-10s" % "iteration" + "".join("%40s" % ("%s - %s" % p) for p in pairs))
        for c in checks:
            row = "%-10d" % c
            for a, b in pairs:
                d = [runs[s]["arms"][a][str(c)][key] - runs[s]["arms"][b][str(c)][key] for s in use
                     if a in runs[s]["arms"] and b in runs[s]["arms"]]
                if len(d) < 2:
                    row += "%40s" % "-"
                    continue
                m = sum(d) / len(d)
                sd = math.sqrt(sum((x - m) ** 2 for x in d) / (len(d) - 1))
                hw = T95.get(len(d) - 1, 2.0) * sd / math.sqrt(len(d))
                row += "%40s" % ("%+.3f [%+.3f, %+.3f] (n=%d; p=%.2f)" % (m, m - hw, m + hw, len(d), sign_test(d)))
            print(row)
    print("\n== wall seconds of training up to the last checkpoint (validation renders excluded), mean ==")
    for a in arms:
        v = [runs[s]["arms"][a][last]["train_wall_s"] for s in use if a in runs[s]["arms"]]
        if v:
            print("%-10s %8.1f s" % (a, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
