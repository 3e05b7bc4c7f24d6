import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd
from bhr_amd import HipRenderer, scenes
W, H = 1920, 1080
sky, tex = scenes.analytic_skybox(1024, 2048), scenes.noisy_disk(416, 2912)
for name, kw in (("default disk 2-15", dict(r_disk_inner=2.0, r_disk_outer=15.0)), ("no disk (2-2.0001)", dict(r_disk_inner=2.0, r_disk_outer=2.0001))):
    for math in ("fast", "strict"):
        r = HipRenderer(W, H, sky, tex, math=math, **kw)
        for comp in (False, True):
            for _ in range(5): r.render_async([6, 0, 0.5], 90, compaction=comp)
            r.timing_reset()
            for _ in range(20): r.render_async([6, 0, 0.5], 90, compaction=comp)
            c = r.counters()
            print(f"{name:22s} {math:6s} {'persist' if comp else 'tile':8s} march {c['march_ms_sum']/c['frames_timed']:.3f} ms  bloom {c['bloom_ms_sum']/c['frames_timed']:.3f} ms steps/ray {c['ray_steps']/W/H:.2f}")
        r.close()
