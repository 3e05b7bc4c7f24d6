"""Per-frame kernel table of a rocprofv3 --kernel-trace run of tools/exp_video.py.
Usage: summarize_video_prof.py <rocprof output dir> <n_frames> [note]"""
import collections, csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
note = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
dur = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("::")[-1].split("(")[0] if "::" in name else name.split("(")[0]
    if "png_plan" in name: short = "png_plan_kernel"
    if "png_scan" in name: short = "png_scan_kernel"
    if "png_encode" in name: short = "png_encode_kernel"
    dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
march = sorted((int(r["Start_Timestamp"]) for r in rows if "march" in r["Kernel_Name"]))
period = (march[-50] - march[50]) / (len(march) - 100) / 1e3 if len(march) > 200 else float("nan")
print(f"# Video loop at 1920x1080 (orbit, lifecycle texture every frame, device PNG): kernels per frame\n")
print(f"{note}  \nunder the profiler: {period:.0f} us between successive march launches ({1e6 / period:.0f} frames/s)\n")
print("| kernel | launches / frame | avg us | us / frame |\n|---|---|---|---|")
tot = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    per_frame = sum(v) / n / 1e3
    if per_frame < 0.5:
        continue
    tot += per_frame
    print(f"| `{k}` | {len(v) / n:.2f} | {sum(v) / len(v) / 1e3:.1f} | {per_frame:.1f} |")
print(f"| sum | | | {tot:.0f} |")
