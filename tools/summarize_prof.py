#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# optional: --traffic <workload> <profiles/march_traffic.json> updates the per-launch HBM bytes of the march from the
# FETCH_SIZE / WRITE_SIZE passes summarised here (no hand-typed numbers; bench.py reads the file into roofline.traffic)
traffic_args = sys.argv[sys.argv.index("--traffic") + 1:sys.argv.index("--traffic") + 3] if "--traffic" in sys.argv else None


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def short(name):
    """'void (anonymous namespace)::march_tile_kernel<true, 0>(BhrMarchArgs)' -> 'march_tile_kernel<true, 0>'"""
    import re
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    m = re.match(r"([A-Za-z_][A-Za-z0-9_:]*)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
st = find("trace", "kernel_stats.csv")
if st:
    print("## kernel durations (--kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(st)):
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    print()

for sub in ("pmc_a", "pmc_b", "pmc_fetch", "pmc_write"):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        meta[k] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
                   r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
    print(f"## {sub}: per-dispatch averages\n")
    names = sorted({c for k in acc for c in acc[k]})
    print("| kernel | vgpr/sgpr/lds | grid | " + " | ".join(names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    for k in acc:
        row = [f"{acc[k][c] / max(cnt[k][c], 1):.4g}" if c in acc[k] else "" for c in names]
        m = meta[k]
        print(f"| {k} | {m[0]}/{m[1]}/{m[2]} | {m[3]}x{m[4]} | " + " | ".join(row) + " |")
    print()


if traffic_args:
    import json
    workload, path = traffic_args                    # workload key, e.g. "fhd/hybrid" (bench.py looks up "<workload>/<math>")
    vals = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(sub, "counter_collection.csv")
        tot, n = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "march" in r["Kernel_Name"]:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
        frames = max(n.values())                     # a hybrid march is two kernels per frame: their bytes add up
        vals[name] = ({k: round(tot[k] / n[k], 1) for k in n}, sum(tot.values()) / frames, frames)
    # FP32 flop the march kernels execute per frame: 64 lanes x (2 x FMA + MUL + ADD wave-instructions), pmc_b pass
    flop = None
    try:
        f = find("pmc_b", "counter_collection.csv")
        tot, n = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "march" in r["Kernel_Name"] and r["Counter_Name"] in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32"):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
        frames_b = max(n.values()) if n else 0
        if frames_b:
            flop = 64.0 * (2.0 * tot["SQ_INSTS_VALU_FMA_F32"] + tot["SQ_INSTS_VALU_MUL_F32"] + tot["SQ_INSTS_VALU_ADD_F32"]) / frames_b
    except Exception:
        flop = None
    tj = json.load(open(path)) if os.path.isfile(path) else {}
    tj[workload] = (vals["FETCH_SIZE"][1] + vals["WRITE_SIZE"][1]) * 1024.0
    tj[f"_{workload}_detail"] = {"kernels_FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"][0], "kernels_WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"][0],
                                 "FETCH_SIZE_KiB_per_frame": vals["FETCH_SIZE"][1], "WRITE_SIZE_KiB_per_frame": vals["WRITE_SIZE"][1],
                                 "frames": vals["FETCH_SIZE"][2],
                                 "profile": f"profiles/{os.path.basename(out).replace('prof_', '')}.md",
                                 "executed_fp32_flop_per_frame": flop,
                                 "commit": os.environ.get("BHR_COMMIT", "unknown")}
    tj["source"] = "tools/summarize_prof.py from the pmc_fetch / pmc_write CSVs of tools/profile.sh; per workload/math: _<key>_detail"
    tj["_note"] = ("HBM bytes per march (all its launches of one frame) from rocprofv3 --pmc (separate passes): (FETCH_SIZE + WRITE_SIZE) KiB x 1024. "
                   "WRITE_SIZE equals the two f32 framebuffers; FETCH_SIZE is uncalibrated for this kernel's 12-16 B gathers "
                   "(MI355X_MICROARCH.md: the x2 correction applies to wide coalesced streams only) and is reported uncorrected.")
    json.dump(tj, open(path, "w"), indent=1)
    print(f"updated {path}: {workload} = {tj[workload]:.0f} B per frame", file=sys.stderr)
