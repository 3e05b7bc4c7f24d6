#!/bin/bash
# kernel durations of the post-pass at the BASELINE sizes: tools/trace_bloom.sh <tag> [exp_bloom args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_bloom_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in fhd 4k 8k_tile 8k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$s -- python3 $ROOT/tools/exp_bloom.py $s split=1 out=u8 "$@" > $OUT/$s.log 2>&1
  echo "== $s"; grep -h "bloom_\|march_tile" $OUT/$s/*/*kernel_stats.csv | awk -F'","' '{printf "%-60s calls %s avg_us %.1f\n", substr($1,2,60), $2, $4/1000}'
done
