#!/bin/bash
# LDS counters of the texture pipeline kernels (background_kernel: is it the bank conflicts?), run on the GPU box:
#   tools/pmc_texture.sh  -> gpurun_out/pmc_texture.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_texture
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python3 $ROOT/tools/exp_bg.py > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 $ROOT/tools/exp_bg.py > $OUT/b.log 2>&1
OUT=$OUT python3 - <<'P'
import os
OUT = os.environ['OUT']
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(OUT + "/*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen: cnt[(k, f)] += 1
n = collections.defaultdict(int)
for (k, f), c in cnt.items(): n[k] = max(n[k], c)
print("| kernel | launches | " + " | ".join(sorted({c for v in tot.values() for c in v})) + " |")
cols = sorted({c for v in tot.values() for c in v})
print("|---|---|" + "---|" * len(cols))
for k, v in sorted(tot.items()):
    if n[k] < 50: continue
    print(f"| `{k}` | {n[k]} | " + " | ".join(f"{v.get(c, 0) / n[k]:.4g}" for c in cols) + " |")
P
