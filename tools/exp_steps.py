"""March time against the number of steps per ray (step_size sweep at fhd): fixed per-ray cost vs per-step cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhr_amd import HipRenderer, scenes
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
for math in ("strict", "fast"):
    for ss in (0.4, 0.2, 0.1, 0.05, 0.025):
        r = HipRenderer(1920, 1080, sky, tex, step_size=ss, math=math)
        for _ in range(4):
            r.render_async([6, 0, 0.5], 90)
        c = r.counters()
        print(f"{math:6s} step {ss:5.3f}: march {c['march_ms']:.3f} ms, {c['ray_steps'] / 2073600:.1f} steps/ray, "
              f"{c['march_ms'] * 1e6 / c['ray_steps'] * 1e3:.3f} ps/ray-step", flush=True)
        r.close()
