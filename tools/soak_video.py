"""Long-run check of the video path: many frames at sd, entity turnover, profile-pool resets, frame sink; reports
device memory before/after and the frame rate.  Usage: soak_video.py [n_frames]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bhr_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tmp = tempfile.mkdtemp(prefix="bhr_soak_")
free0, total = torch.cuda.mem_get_info(0)
r, _, _, _ = drivers.make_renderer(640, 360, [6, 0, 0.5], 90, n_stars=500)
free1, _ = torch.cuda.mem_get_info(0)
t0 = time.perf_counter()
drivers.render_video(r, 640, 360, n_frames=n, fps=30, output_path=os.path.join(tmp, "v.mp4"), fov=90,
                     static_cam_pos=[6, 0, 0.5], orbit=True, assemble=False)
dt = time.perf_counter() - t0
free2, _ = torch.cuda.mem_get_info(0)
frames = len([f for f in os.listdir(drivers._frames_dir(os.path.join(tmp, "v.mp4"))) if f.endswith(".png")])
print(f"{n} frames in {dt:.1f} s = {n / dt:.0f} fps; {frames} PNG files; device memory used by the renderer "
      f"{(free0 - free1) / 2**20:.0f} MiB after set-up, {(free0 - free2) / 2**20:.0f} MiB after the run", flush=True)
r.close()
free3, _ = torch.cuda.mem_get_info(0)
print(f"after close: {(free0 - free3) / 2**20:.0f} MiB still held (torch context)", flush=True)
shutil.rmtree(tmp)
