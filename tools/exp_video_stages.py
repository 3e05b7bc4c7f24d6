"""Where a frame of the video driver's loop goes (configs[4], fhd, hybrid march, device PNG): host time of each call of the
loop body, the loop with stages removed (no PNG sink / no texture work / no march), and the loop as it is.
Usage: python tools/exp_video_stages.py [n_frames]"""
import os, sys, time, shutil, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import drivers
from bhr_amd.output import FrameSink, DEVICE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tmp = tempfile.mkdtemp(prefix="bhr_video_")
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math="hybrid")
n_r, n_phi = r.dtex_h, r.dtex_w
factories = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
res = {}


def loop(tag, texture=True, march=True, sink_on=True, frames=n):
    sink = FrameSink(r, slots=0, workers=4, level=DEVICE) if sink_on else None
    acc = {"advance": 0.0, "render": 0.0, "submit": 0.0}
    for f in range(20):                                      # warm-up
        drivers.advance_lifecycle_frame(r, factories, f * 0.1, 0.1, recompute_stats=(f % 60 == 0), compose=True)
        r.render_async([6, 0, 0.5], 90)
    r.sync()
    t_all = time.perf_counter()
    for f in range(frames):
        t0 = time.perf_counter()
        if texture:
            drivers.advance_lifecycle_frame(r, factories, f * 0.1, 0.1, recompute_stats=(f % 60 == 0), compose=True)
        t1 = time.perf_counter()
        if march:
            r.render_async(drivers.orbit_position([6, 0, 0.5], f, frames, 360.0), 90, frame=0)
        t2 = time.perf_counter()
        if sink is not None:
            sink.submit(os.path.join(tmp, f"frame_{f:04d}.png"))
            if (f + 1) % 50 == 0:
                sink.drain()
        t3 = time.perf_counter()
        acc["advance"] += t1 - t0; acc["render"] += t2 - t1; acc["submit"] += t3 - t2
    if sink is not None:
        sink.drain(); sink.close()
    r.sync()
    dt = time.perf_counter() - t_all
    res[tag] = {"fps": frames / dt, "ms_per_frame": dt / frames * 1e3, **{k + "_host_ms": v / frames * 1e3 for k, v in acc.items()}}
    print(tag, json.dumps(res[tag]), flush=True)


loop("full")
loop("no_sink", sink_on=False)
loop("no_texture", texture=False)
loop("march_only", texture=False, sink_on=False)
loop("texture_only", march=False, sink_on=False)
loop("full_again")
r.close()
shutil.rmtree(tmp)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "video_stages.json"), "w"), indent=1)
