"""A/B of one library option on the post-pass time: python tools/exp_bloom_ab.py <option> <v0> <v1> [sizes...]  (alternating, 4 rounds)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bhr_amd import HipRenderer, scenes
from exp_bloom import SIZES, KW

opt, v0, v1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
for name in sys.argv[4:] or ["fhd", "4k", "8k_tile", "8k"]:
    W, H, rows = SIZES[name]
    r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, rows=rows, outputs="u8", options={"bloom_split": 1}, **dict(KW, step_size=0.3 if W > 4000 else 0.1))
    res = {v0: [], v1: []}
    for rnd in range(5):
        for v in (v0, v1):
            r.set_option(opt, v)
            for _ in range(3):
                r.render_async([6, 0, 0.5], 90)
            r.timing_reset()
            for _ in range(20):
                r.render_async([6, 0, 0.5], 90)
            c = r.counters()
            res[v].append(c["bloom_ms_sum"] / c["frames_timed"])
    print(name, {k: [round(x, 4) for x in v[1:]] for k, v in res.items()}, "medians", {k: round(float(np.median(v[1:])), 4) for k, v in res.items()}, flush=True)
    r.close()
