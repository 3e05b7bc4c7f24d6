#!/bin/bash
# tools/sweep_env.sh "ENV1=a ENV2=b" "ENV1=c" ... : bench.py (fhd strict) under each environment, one and two frames in flight
for e in "$@"; do for sl in 1 2; do
  env $e python bench.py --no-cpu-baseline --no-other-math --tile-workload none --frame-slots $sl ${BENCH_ARGS} 2>/dev/null > /tmp/sweep.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(sys.argv[1], "| slots", d["config"]["frame_slots"], "fps", round(d["fps"], 1), "march_ms", round(d["kernel_ms"]["march"], 4))
PY
done; done
