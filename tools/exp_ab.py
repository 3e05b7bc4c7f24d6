"""A/B timing of library variants: tools/exp_ab.py lib1.so lib2.so ...  (fhd default scene, fast math)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import bhr_amd
from bhr_amd import HipRenderer, scenes
from oracle import oracle as O
W, H = 1920, 1080
sky, tex = scenes.analytic_skybox(1024, 2048), scenes.noisy_disk(416, 2912)
res = {}
for aa in ("disabled", "lod_radius"):
    r = HipRenderer(W, H, sky, tex, math="fast", anti_alias=aa)
    for comp in (False, True):
        for _ in range(5): r.render_async([6, 0, 0.5], 90, compaction=comp)
        r.timing_reset()
        for _ in range(30): r.render_async([6, 0, 0.5], 90, compaction=comp)
        c = r.counters()
        res[f"{aa[:3]}/{'pers' if comp else 'tile'}"] = round(c['march_ms_sum']/c['frames_timed'], 4)
    r.close()
# accuracy of this build on the small default scene
s = scenes.SCENES["default"]
sk, tx = scenes.analytic_skybox(), scenes.noisy_disk()
ref = O.OracleRenderer(s["width"], s["height"], sk, tx, **s["kw"]).render(s["cam_pos"], s["fov"])
h = HipRenderer(s["width"], s["height"], sk, tx, math="fast", **s["kw"])
h.render_async(s["cam_pos"], s["fov"], compaction=False)
out = h.read_layer(0)
res["rmse"] = float(np.sqrt(np.mean((out.astype(np.float64)-ref)**2)))
print(res)
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, BHR_HIP_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
