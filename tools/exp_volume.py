"""Cost of the Disk V2 sources at fhd: texture / analytic surface / finite-thickness volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhr_amd import HipRenderer, scenes, disk_v2 as dv
P = dv.DiskV2Params()
for math in ("strict", "fast"):
    r = HipRenderer(1920, 1080, scenes.analytic_skybox(), scenes.noisy_disk(), r_disk_inner=P.r_in, r_disk_outer=P.r_out, math=math)
    for name, setup in (("texture", lambda: r.use_disk_v2(None)), ("dv2 surface", lambda: r.use_disk_v2(P)),
                        ("dv2 volume x2", lambda: r.use_disk_v2(P, volume=True, substeps=2)),
                        ("dv2 volume x4", lambda: r.use_disk_v2(P, volume=True, substeps=4))):
        setup()
        for cam in ([6, 0, 0.5], [9, 0, 0.6]):
            for _ in range(3):
                r.render_async(cam, 90)
            c = r.counters()
            print(f"{math:6s} {name:14s} cam {cam}: march {c['march_ms']:.3f} ms, {c['ray_steps'] / 1e6:.1f} M ray-steps", flush=True)
    r.close()
