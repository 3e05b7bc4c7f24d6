"""cProfile of the video driver's loop body (configs[4], fhd, hybrid, device PNG): which calls hold the host.
Usage: python tools/exp_video_profile.py [n_frames]"""
import cProfile, pstats, os, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhr_amd import drivers
from bhr_amd.output import FrameSink, DEVICE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tmp = tempfile.mkdtemp(prefix="bhr_video_")
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math="hybrid")
factories = drivers.init_lifecycle_system(r, r.dtex_h, r.dtex_w, seed=42)
sink = FrameSink(r, slots=0, workers=4, level=DEVICE)


def body(frames, f0=0):
    for f in range(f0, f0 + frames):
        drivers.advance_lifecycle_frame(r, factories, f * 0.1, 0.1, recompute_stats=(f % 60 == 0), compose=True)
        r.render_async(drivers.orbit_position([6, 0, 0.5], f, 3600, 360.0), 90, frame=0)
        sink.submit(os.path.join(tmp, f"frame_{f:04d}.png"))
        if (f + 1) % 50 == 0:
            sink.drain()


body(300)                      # clocks, caches
t0 = time.perf_counter()
body(n, 300)
sink.drain()
print(f"un-profiled: {n / (time.perf_counter() - t0):.0f} fps")
pr = cProfile.Profile()
pr.enable()
body(n, 300 + n)
sink.drain()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
sink.close(); r.close(); shutil.rmtree(tmp)
