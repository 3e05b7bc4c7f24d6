#!/bin/bash
# One PMC pass restricted to the march kernel: LDS use and conflicts beside the issue counters.
#   tools/pmc_march.sh [bench args, default: fhd]   (run on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_march
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-math --tile-workload none --video-frames 0 --frame-slots 1 "$@" > $OUT/a.log 2>&1
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["OUT"] + "/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if "march" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(acc): print(k, f"{acc[k] / cnt[k]:.4g}")
PY
