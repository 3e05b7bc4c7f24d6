#!/bin/bash
# PMC passes over the split post-pass kernels: tools/pmc_bloom.sh <tag> <exp_bloom args...>   (run on the GPU box through gpurun)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bloom_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/exp_bloom.py $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $CMD > $OUT/write.log 2>&1
cd $ROOT && python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
