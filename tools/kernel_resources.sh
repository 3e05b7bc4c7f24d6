#!/bin/bash
# Registers / spills / LDS of every kernel in the built objects, from the code-object metadata (no GPU needed).
# usage: tools/kernel_resources.sh [object-name-filter]
set -e
OBJ=/root/repo/black-hole-renderer_amd/lib/obj
T=$(mktemp -d)
for o in $OBJ/*${1}*.o; do
  b=$(basename $o)
  cp $o $T/$b
  (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $b >/dev/null 2>&1 || true)
  co=$(ls $T/$b.*gfx950 2>/dev/null | head -1)
  [ -z "$co" ] && continue
  echo "== $b"
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | awk '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2}
    /\.agpr_count:/ {a=$2}
    /\.vgpr_spill_count:/ {sp=$2}
    /\.sgpr_count:/ {s=$2}
    /\.private_segment_fixed_size:/ {pr=$2}
    /\.group_segment_fixed_size:/ {l=$2}
    /\.wavefront_size:/ {printf "%-70s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s\n", name, v, a, s, sp, pr, l}'
done
rm -rf $T
