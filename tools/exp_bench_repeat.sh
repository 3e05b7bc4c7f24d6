#!/bin/bash
# tools/exp_bench_repeat.sh N [bench args]: the bench headline N times in fresh processes, every JSON line kept (gpurun_out/repeat/<k>.json)
N=$1; shift
mkdir -p gpurun_out/repeat
for k in $(seq $N); do
  python bench.py --no-cpu-baseline --no-other-math --tile-workload none --no-config2 --video-frames 0 --tile-tail-tiles 0 "$@" 2>/dev/null > gpurun_out/repeat/$k.json
  python - gpurun_out/repeat/$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
t = d.get("kernel_ms_in_timed_region", {})
print(sys.argv[1], "fps", round(d["fps"], 1), "calib", d.get("stream_calibration", {}).get("kept"), "busy/span", round(t.get("march_busy_ms", 0), 1), round(t.get("span_ms", 0), 1),
      "march_in_region", round(t.get("march", 0), 3), "region", {k: round(v, 2) for k, v in d.get("timed_region_ms", {}).items()} if "timed_region_ms" in d else "", flush=True)
PY
done
