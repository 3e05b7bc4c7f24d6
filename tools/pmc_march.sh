set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc4k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -- python3 $ROOT/bench.py --workload 4k --steps 10 --warmup 2 --no-cpu-baseline --no-other-math > $OUT/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    if "march" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]["v"]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for k in acc: print(k, acc[k]["v"]/cnt[k])
PY
