"""When does a two-slot context drop to its slow state?  One fresh process: the bench scene, spin-up (with the stream calibration),
then blocks of 100 frames timed one by one (a sync between blocks), with or without a torch.cuda.synchronize() between them.
usage: python tools/exp_stream_state.py [blocks] [torch]"""
import json, os, sys, time
if "torch" in sys.argv:
    import torch            # before the library, as bench.py has it (one HIP runtime in the process)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bhr_amd import workloads

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
use_torch = "torch" in sys.argv
wl = bench.WORKLOADS["fhd"]
r, _, _, _ = workloads.make_scene(wl, math="hybrid", frame_slots=2)
if use_torch:
    torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 0.3:
    for _ in range(8):
        r.render_async(wl["cam_pos"], wl["fov"])
    r.sync()
    n += 8
out = []
for b in range(blocks):
    r.sync()
    if use_torch and b % 3 == 2:
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        r.render_async(wl["cam_pos"], wl["fov"])
    r.sync()
    out.append(round(100 / (time.perf_counter() - t)))
print(json.dumps({"calib": r.stream_calibration(), "blocks_fps": out, "torch": use_torch}), flush=True)
r.close()
