#!/bin/bash
for r in 1 2 3; do for lib in "$@"; do
  BHR_HIP_LIBRARY=$(realpath $lib) python bench.py --no-cpu-baseline --tile-workload none --video-frames 0 --steps 40 2>/dev/null > /tmp/ab.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
c = d["config2_4k"]
print(sys.argv[1][-30:], "| config2 fps", round(c["fps_one_frame_at_a_time"], 1), "march_ms", round(c["kernel_ms"]["march"], 4), flush=True)
PY
done; done
