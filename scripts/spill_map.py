"""Where the register spills of the fp16-piece forward kernels sit ("114 SGPR spills + 10 VGPR spills in the dominant
kernel"): compiles mlp_f16w.hip to gfx950 assembly with the product's flags and, per kernel, maps every spill instruction
(v_writelane / v_readlane = an SGPR parked in / fetched from a VGPR lane; scratch_store / scratch_load = a VGPR spilled to memory)
onto the loop nest (backward branches) and onto the blocks that hold the multiply loops (>= 18 MFMAs):
    python scripts/spill_map.py > profiles/rNN_spill_map.txt"""
import os
import re
import subprocess
import tempfile

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-pytorch_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm "
         "-pragma-unroll-threshold=4000000 -S --cuda-device-only").split()
SPILL = ("v_writelane", "v_readlane", "scratch_store", "scratch_load")
MFMA_BLOCK = 18


def demangle(sym):
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def kernels(lines):
    heads = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^(_Z\w+):", ln)] if m]
    for (a, sym), (b, _) in zip(heads, heads[1:] + [(len(lines), "")]):
        end = next((i for i in range(a, b) if lines[i].strip().startswith("s_endpgm")), b)
        yield demangle(sym), a, end


def analyse(lines, a, b):
    blocks, cur, lab_line, branches = [], ["entry", a, {}], {}, []
    for i in range(a, b):
        ln = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append(cur)
            cur = [m.group(1), i, {}]
            lab_line[m.group(1)] = i
            continue
        op = ln.split()[0] if ln and not ln.startswith(";") else ""
        for key in SPILL + ("v_mfma",):
            if op.startswith(key):
                cur[2][key] = cur[2].get(key, 0) + 1
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
        if m:
            branches.append((i, m.group(1)))
    blocks.append(cur)
    loops = [(lab_line[t], i) for i, t in branches if t in lab_line and lab_line[t] <= i]
    return blocks, loops


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "mlp_f16w.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["mlp_f16w.hip", "-o", out], cwd=HERE, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    print("# python scripts/spill_map.py   (mlp_f16w.hip -> gfx950 assembly with the product's flags; counts are INSTRUCTIONS in the")
    print("# text, not executions.  'prologue/epilogue' = outside every loop; 'phase code' = inside the tile / layer loops but in a")
    print("# block without a multiply loop; 'multiply blocks' = blocks with >= %d MFMAs.  v_readlane is also what the kernels' own" % MFMA_BLOCK)
    print("# wave-wide reductions use: the data-gradient kernels spill nothing and still show 3-5)")
    print("%-46s %6s | %-23s | %-23s | %-32s" % ("kernel", "MFMAs", "prologue/epilogue", "phase code", "multiply blocks (n; worst block)"))
    print("%-46s %6s | %-23s | %-23s | %-32s" % ("", "", "wl / rl / sst / sld", "wl / rl / sst / sld", "wl / rl / sst / sld"))
    for name, a, b in kernels(lines):
        blocks, loops = analyse(lines, a, b)
        where = {"pro": [0] * 4, "phase": [0] * 4, "mul": [0] * 4}
        nmul, worst, mfmas = 0, 0, 0
        for _, i, d in blocks:
            depth = sum(1 for lo, hi in loops if lo <= i + 1 <= hi)
            mf = d.get("v_mfma", 0)
            mfmas += mf
            k = "mul" if mf >= MFMA_BLOCK else ("pro" if depth == 0 else "phase")
            cnt = [d.get(s, 0) for s in SPILL]
            where[k] = [x + y for x, y in zip(where[k], cnt)]
            if k == "mul":
                nmul += 1
                worst = max(worst, sum(cnt))
        fmt = lambda v: "%3d / %3d / %3d / %3d" % tuple(v)
        print("%-46s %6d | %-23s | %-23s | %s  (%d; %d)" % (name[:46], mfmas, fmt(where["pro"]), fmt(where["phase"]), fmt(where["mul"]),
                                                           nmul, worst))


if __name__ == "__main__":
    main()
