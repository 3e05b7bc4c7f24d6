"""Video driver throughput at fhd (configs[4]): frames/s end to end with the frame sink, and the
time of each stage.  Usage: python tools/exp_video.py [n_frames] [png_level, -1 = device encoder] [workers] [math = hybrid]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhr_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
math = sys.argv[4] if len(sys.argv) > 4 else "hybrid"
print("cpus", len(os.sched_getaffinity(0)), flush=True)
tmp = tempfile.mkdtemp(prefix="bhr_video_")
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math=math)
t0 = time.perf_counter()
drivers.render_video(r, 1920, 1080, n_frames=n, fps=30, output_path=os.path.join(tmp, "v.mp4"), fov=90,
                     static_cam_pos=[6, 0, 0.5], orbit=True, assemble=False, png_level=level, sink_workers=workers)
dt = time.perf_counter() - t0
print(f"render_video: {n} frames in {dt:.2f} s = {n / dt:.1f} fps (includes lifecycle init)", flush=True)
# the same frames through PIL on 2 threads, the reference's arrangement
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
r.render_async([6, 0, 0.5], 90)
u8 = r.read_final_u8()
t0 = time.perf_counter()
with ThreadPoolExecutor(2) as pool:
    list(pool.map(lambda k: Image.fromarray(u8).save(os.path.join(tmp, f"pil{k}.png")), range(16)))
dt = time.perf_counter() - t0
print(f"PIL 2 threads: {16 / dt:.1f} fps ({os.path.getsize(os.path.join(tmp, 'pil0.png')) / 1e6:.2f} MB/frame)")
from bhr_amd.output import png_encode
for lv, th in ((1, 1), (6, 1), (1, 8), (6, 8)):
    t0 = time.perf_counter(); d = png_encode(u8, lv, th); dt = time.perf_counter() - t0
    print(f"png_encode level {lv} threads {th}: {dt * 1e3:.1f} ms {len(d) / 1e6:.2f} MB")
from bhr_amd.output import png_encode_device
png_encode_device(r)
t0 = time.perf_counter()
for _ in range(20):
    d = png_encode_device(r)
dt = (time.perf_counter() - t0) / 20
print(f"png_encode_device (quantise + 3 launches + fetch, synchronous): {dt * 1e3:.2f} ms {len(d) / 1e6:.2f} MB")
r.close()
shutil.rmtree(tmp)
