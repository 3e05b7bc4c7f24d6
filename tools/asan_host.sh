#!/bin/bash
# Host-side AddressSanitizer build of libbhr_hip.so (device code is untouched: GPU ASAN is not available on
# this pool) and the CPU tests that execute library code without a GPU: the PNG encoder and the ABI surface.
#   tools/asan_host.sh            -> builds into /tmp/bhr_asan, runs tests/test_png.py tests/test_abi.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/bhr_asan
mkdir -p $OUT
cd $ROOT/black-hole-renderer_amd/csrc
make -s
for f in api output png_device flare lifecycle api_disk_v2 disk_v2 bloom texture skyglow; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -DBHR_BUILD -fsanitize=address \
      -fno-omit-frame-pointer -Wno-option-ignored -c $f.hip -o $OUT/$f.o &
done
wait
cp ../lib/obj/march.o ../lib/obj/march_strict.o ../lib/obj/march_strict_ilp.o $OUT/      # device-heavy objects: no host code worth instrumenting
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -o $OUT/libbhr_hip.so $OUT/*.o -lz -lpthread
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 BHR_HIP_LIBRARY=$OUT/libbhr_hip.so LD_PRELOAD=$ASAN python -m pytest tests/test_png.py tests/test_abi.py tests/test_png_device.py -q -m "not gpu"
