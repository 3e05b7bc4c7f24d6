#!/bin/bash
# tools/exp_bench_repeat.sh N [bench args]: the bench headline N times in fresh processes, every JSON line kept (gpurun_out/repeat/<k>.json);
# per run: fps, the calibration's choice, the frames' own event span against the host's timed region (renderer_sync = the wait's wake-up)
N=$1; shift
mkdir -p gpurun_out/repeat
for k in $(seq $N); do
  python bench.py --no-cpu-baseline --no-other-math --tile-workload none --no-config2 --video-frames 0 --tile-tail-tiles 0 "$@" 2>/dev/null > gpurun_out/repeat/$k.json
  python - gpurun_out/repeat/$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
t = d.get("kernel_ms_in_timed_region", {})
r = d.get("timed_region_ms", {})
print(sys.argv[1], "fps", round(d["fps"], 1), "calib", d.get("stream_calibration", {}).get("kept"), "event span", round(t.get("span_ms", 0), 2),
      "host total", round(r.get("total", 0), 2), "submit", round(r.get("host_submit", 0), 2), "sync", round(r.get("renderer_sync", 0), 2), flush=True)
PY
done
