"""Per-kernel averages of rocprofv3 --pmc CSVs (one directory per pass) + the --stats durations: tools/summarize_pmc.py <dir> [kernel-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict

def clean(name):
    """'void (anonymous namespace)::bloom_h_split_kernel<5>((anonymous namespace)::HSplitArgs)' -> 'bloom_h_split_kernel<5>'"""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    return n.split("(")[0]


def main():
    root = sys.argv[1]
    filt = sys.argv[2:] or ["bloom_"]
    dur = {}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k in sorted(acc):
        if not any(s in k for s in filt):
            continue
        short = clean(k)
        d = next((v for n, v in dur.items() if clean(n) == short), None)
        print(f"== {short}  calls/avg_us = {d}")
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print(f"   {c:28s} {s / n:16.1f}   (n={n})")

if __name__ == "__main__":
    main()
