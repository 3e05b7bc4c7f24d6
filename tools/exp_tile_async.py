import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
os.environ["BHR_TILE_DEVICES"] = ",".join(["0"] * 8)
r = bench.tile_leg(bench.WORKLOADS["8k"], 8, 20, math="hybrid")
print(json.dumps({k: r[k] for k in ("ms_per_frame", "ms_per_frame_each_waited_for", "fps", "tile_ms")}))
