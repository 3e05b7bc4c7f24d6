#!/bin/bash
# per-kernel durations of the device PNG encoder at fhd / 4k / 8k (run on the GPU box): tools/exp_png_prof.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_png
rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/exp_png.py fhd 4k 8k > $OUT.log 2>&1
grep -v "^[EW]2026" $OUT.log | tail -4
python3 - <<'P'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_png/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "png_" in r["Kernel_Name"] or "quantize" in r["Kernel_Name"]:
        d[(r["Kernel_Name"].split("::")[1].split("(")[0], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = sorted(v)
    print(k, len(v), "median us", v[len(v) // 2] / 1e3)
P
