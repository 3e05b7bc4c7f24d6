"""cProfile of the video driver's host side at fhd (configs[4]) with the device PNG encoder: where a frame's
1.7 ms of host time goes.  Usage: python tools/prof_video.py [n_frames]"""
import cProfile, pstats, os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhr_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tmp = tempfile.mkdtemp(prefix="bhr_video_")
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
drivers.render_video(r, 1920, 1080, n_frames=n, fps=30, output_path=os.path.join(tmp, "v.mp4"), fov=90,
                     static_cam_pos=[6, 0, 0.5], orbit=True, assemble=False)
pr.disable()
print(f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms/frame (profiled)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
r.close()
shutil.rmtree(tmp)
