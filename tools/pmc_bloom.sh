#!/bin/bash
# Kernel durations + issue / LDS counters of the bloom passes at 8k for one H,V variant pair (run on the GPU box):
#   tools/pmc_bloom.sh <tag> <H variant> <V variant>
set -e
TAG=$1; HV=$2; VV=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bloom_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/exp_bloom.py --sizes 8k --quick $HV,$VV"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/c --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $CMD > $OUT/c.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $OUT/b --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS -- $CMD > $OUT/b.log 2>&1
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bloom" in r["Name"]:
            print("stats", r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
for p in ("a", "b", "c"):
    fs = glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "bloom_h" in r["Kernel_Name"] or "bloom_v" in r["Kernel_Name"]:
            key = ("H" if "bloom_h" in r["Kernel_Name"] else "V", r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); cnt[key] += 1
    for k in sorted(acc):
        print("pmc", k[0], k[1], f"{acc[k] / cnt[k]:.5g}")
PY
