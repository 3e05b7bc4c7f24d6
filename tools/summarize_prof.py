#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


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
