#!/bin/bash
# tools/exp_lib_ab.sh lib1.so lib2.so ... : bench.py (fhd hybrid, the headline) under each build of the library (BHR_HIP_LIBRARY),
# alternating, ROUNDS rounds; prints fps (two frames in flight) and the isolated march bracket
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do for lib in "$@"; do
  BHR_HIP_LIBRARY=$(realpath $lib) python bench.py --no-cpu-baseline --no-other-math --tile-workload none ${BENCH_ARGS} 2>/dev/null > /tmp/ab.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "| fps", round(d["fps"], 1), "march_ms", round(d["kernel_ms"]["march"], 4), "post_ms", round(d["kernel_ms"]["bloom_and_combine"], 4), "vgprs", d["kernel_ms"].get("march_vgprs"), "calib", d.get("stream_calibration", {}).get("kept"), d.get("stream_calibration", {}).get("candidates_fps"), "share", d.get("stream_map"), flush=True)
PY
done; done
