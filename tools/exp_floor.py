"""Where does the march's fixed cost come from?  Coarse steps (few iterations) at several frame sizes and with
the disk / sky work removed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhr_amd import HipRenderer, scenes
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
def run(tag, w, h, math="strict", **kw):
    r = HipRenderer(w, h, sky, tex, math=math, **kw)
    for _ in range(4):
        r.render_async([6, 0, 0.5], 90)
    c = r.counters()
    print(f"{tag:28s} {w}x{h} {math:6s}: march {c['march_ms']:.3f} ms, {c['ray_steps'] / (w * h):.1f} steps/ray", flush=True)
    r.close()
for (w, h) in ((640, 360), (1920, 1080), (3840, 2160)):
    run("step 0.4", w, h, step_size=0.4)
run("step 0.4, no disk in reach", 1920, 1080, step_size=0.4, r_disk_inner=200.0, r_disk_outer=201.0)
run("step 0.4, r_max 2 (escape fast)", 1920, 1080, step_size=0.4, r_max=2.0)
run("step 0.1", 1920, 1080, step_size=0.1)
run("step 0.1, no disk in reach", 1920, 1080, step_size=0.1, r_disk_inner=200.0, r_disk_outer=201.0)
run("step 0.1 fast", 1920, 1080, math="fast", step_size=0.1)
run("step 0.1 fast, no disk", 1920, 1080, math="fast", step_size=0.1, r_disk_inner=200.0, r_disk_outer=201.0)
