#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box through gpurun):
#   tools/profile.sh <tag> [bench args...]
# 1. --kernel-trace --stats       per-kernel durations
# 2..5. --pmc passes (own runs)  SQ issue/occupancy counters, FETCH_SIZE, WRITE_SIZE + L2 hit/miss
# Outputs: gpurun_out/prof_<tag>/{trace,pmc_a,pmc_b,pmc_fetch,pmc_write}/..., summary by tools/summarize_prof.py
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --frame-slots 1: one frame at a time on one stream, so that a kernel's duration is its own (bench.py's default
# keeps two frames in flight; its kernel_ms / roofline come from such isolated launches as well)
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --spin-up-ms 0 --no-cpu-baseline --no-other-math --tile-workload none --frame-slots 1 $@"
# the duration pass runs bench.py's DEFAULT step counts (what the driver runs): 23 launches are not enough for the
# clocks to settle and read 4 % high; the counter passes only need a few launches
TRACE="python3 $ROOT/bench.py --no-cpu-baseline --no-other-math --tile-workload none --frame-slots 1 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -- $BENCH > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_b --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE -- $BENCH > $OUT/pmc_b.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_fetch --pmc FETCH_SIZE -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $BENCH > $OUT/pmc_write.log 2>&1
# BHR_WORKLOAD=<fhd|4k|8k> also refreshes profiles/march_traffic.json (copy it back from gpurun_out/)
cd $ROOT && python3 tools/summarize_prof.py $OUT ${BHR_WORKLOAD:+--traffic $BHR_WORKLOAD $ROOT/gpurun_out/march_traffic.json} > $OUT/summary.md 2>$OUT/summary.err || true
tail -60 $OUT/summary.md
