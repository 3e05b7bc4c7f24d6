#!/bin/bash
# tools/exp_env_ab.sh "ENV=a" "ENV=b" ... : the bench headline under each environment, alternating, ROUNDS rounds (fresh processes)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --no-other-math --tile-workload none --no-config2 --video-frames 0 2>/dev/null > /tmp/ab.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "| fps", round(d["fps"], 1), "march_ms", round(d["kernel_ms"]["march"], 4), "calib", d.get("stream_calibration", {}).get("kept"), flush=True)
PY
done; done
