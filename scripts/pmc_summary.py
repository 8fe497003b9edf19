"""Aggregate rocprofv3 --pmc passes (scripts/gpu_pmc.sh) per kernel and launch size into one table.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so the
x2-corrected column is the one to compare with byte counts (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys, collections

root = sys.argv[1]


def load(pattern):
    files = glob.glob(os.path.join(root, pattern), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def short(name):
    for k in ("k_mlp_fwd16", "k_mlp_dgrad16", "k_mlp_fwd_f16x3", "k_mlp_dgrad_f16x3", "k_mlp_fwd", "k_mlp_dgrad",
              "k_wgrad_reduce", "k_wgrad_f16x3", "k_wgrad", "k_bwd64r_reduce", "k_bwd64r", "k_fwd64r"):
        if k in name:
            return k
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for sub in ("sq", "fetch", "write"):
    for r in load("%s/**/*counter_collection.csv" % sub):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = (k, int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
    for r in load("%s/**/*kernel_trace.csv" % sub):
        k = short(r["Kernel_Name"])
        if k and sub == "sq":
            g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
            dur[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)

summary = {}
print("%-16s %9s %8s %9s %9s %9s %9s %8s   %s" % ("kernel", "grid", "dur_ms", "mfma_util", "wait_any", "wait_inst", "active", "clk_GHz",
                                                    "HBM traffic per launch"))
for key in sorted(acc):
    a, c = acc[key], cnt[key]
    avg = lambda n: a[n] / c[n] if c[n] else float("nan")
    d = sum(dur[key]) / len(dur[key]) if dur.get(key) else float("nan")
    wc = avg("SQ_WAVE_CYCLES")
    # SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves; MFMA busy counts cycles summed over SIMDs
    gui = avg("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
    mfma = avg("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 256 * 4) if gui == gui and gui > 0 else float("nan")
    fetch, write = avg("FETCH_SIZE") * 1024 / 1e9, avg("WRITE_SIZE") * 1024 / 1e9
    print("%-16s %9d %8.3f %9.3f %9.3f %9.3f %9.3f %8.2f   FETCH %.2f GB (x2 corr %.2f)  WRITE %.2f GB" % (
        key[0], key[1], d, mfma, avg("SQ_WAIT_ANY") / wc if wc else float("nan"), avg("SQ_WAIT_INST_ANY") / wc if wc else float("nan"),
        avg("SQ_ACTIVE_INST_ANY") / wc if wc else float("nan"), gui / (d * 1e6) if d == d and d > 0 else float("nan"), fetch, 2 * fetch, write))
    summary["%s/%d" % key] = dict(kernel=key[0], grid=key[1], dur_ms=d, mfma_util=mfma, clk_ghz=gui / (d * 1e6) if d == d and d > 0 else None,
                                  fetch_gb_raw=fetch, fetch_gb_x2=2 * fetch, write_gb=write)
if len(sys.argv) > 2:
    # stamp: the fingerprint of the kernel sources these counters were measured on (bench.py refuses a summary whose stamp is
    # not the one of the library it runs: lib_sources_sha16)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
    try:
        bm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bm)
        summary["_lib_sources_sha16"] = bm.lib_sources_sha16()
    except Exception as e:
        summary["_lib_sources_sha16"] = "unavailable: %r" % (e,)
    json.dump(summary, open(sys.argv[2], "w"), indent=1, sort_keys=True)
