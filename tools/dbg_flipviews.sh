for k in 663 920 148 788 696 690; do
  for lib in lib lib_dbg_0.005 lib_dbg_0.01 lib_dbg; do
    BHR_HIP_LIBRARY=$PWD/black-hole-renderer_amd/$lib/libbhr_hip.so timeout -k 10 120 python tools/dbg_hybrid_view.py $k 1200 5 512 320 2>&1 | grep "^hybrid  " | sed "s/^hybrid  */view $k $lib: /" | cut -c1-120
  done
done
