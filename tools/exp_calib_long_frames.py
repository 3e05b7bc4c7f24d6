import time, sys
sys.path.insert(0, '/root/repo')
import bench
from bhr_amd import workloads
wl = bench.WORKLOADS["8k"]
r, _, _, _ = workloads.make_scene(wl, math="hybrid", frame_slots=2)
t = time.perf_counter()
for k in range(12):
    r.render_async(wl["cam_pos"], wl["fov"])
r.sync()
print("8k two-slot: 12 frames incl. calibration decision", round(time.perf_counter() - t, 3), "s", r.stream_calibration())
r.close()
