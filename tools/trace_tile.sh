#!/bin/bash
# kernel timeline of row blocks rendered alone: tools/trace_tile.sh <tag> <tile> [<tile> ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_tile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t$k -- python3 $ROOT/tools/trace_tile.py $k > $OUT/t$k.log 2>&1
  echo "== tile $k"; tail -1 $OUT/t$k.log; python3 $ROOT/tools/trace_tile.py --timeline $OUT/t$k 2 | tee $OUT/t$k.timeline.txt
done
