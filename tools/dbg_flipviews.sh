#!/bin/bash
# The six views of tools/fuzz_hybrid.py 1200 5 --size 512 320 that flipped pixels, under libraries built with wider mip-level
# guards (run on the GPU box).  Build the variants first (guards are overridable at build time, csrc/bhr_internal.h):
#   for g in 0.005 0.01; do make -C black-hole-renderer_amd/csrc OUT=../lib_dbg_$g OBJ=../lib_dbg_$g/obj EXTRA="-DBHR_LOD_GUARD=${g}f"; done
#   make -C black-hole-renderer_amd/csrc OUT=../lib_dbg OBJ=../lib_dbg/obj EXTRA="-DBHR_LOD_GUARD=0.02f"
# Result (round 3, guard 2e-3 in `lib`): 663 920 148 690 clean from 5e-3, 788 from 1e-2, 696 keeps one pixel at any width.
for k in 663 920 148 788 696 690; do
  for lib in lib lib_dbg_0.005 lib_dbg_0.01 lib_dbg; do
    BHR_HIP_LIBRARY=$PWD/black-hole-renderer_amd/$lib/libbhr_hip.so timeout -k 10 120 python tools/dbg_hybrid_view.py $k 1200 5 512 320 2>&1 | grep "^hybrid  " | sed "s/^hybrid  */view $k $lib: /" | cut -c1-120
  done
done
