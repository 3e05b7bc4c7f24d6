#!/usr/bin/env python3
"""BHR_MIP_LDS (coarse mip levels of the disk texture staged in LDS, csrc/march.hip: march_tile_mipstaged_kernel) against
the plain fast anti-aliased march at 4k, textures small enough for their levels 2-3 / 3 to fit 44 KB.
usage: python tools/exp_mip_lds.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import HipRenderer, scenes
KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=25.0, anti_alias="lod_radius", aa_strength=2.0)
W, H = 3840, 2160
sky = scenes.analytic_skybox()
for tex_hw in ((128, 256), (128, 512), (256, 1024)):
    tex = scenes.noisy_disk(*tex_hw)
    out = {}
    for on in ("0", "1"):
        os.environ["BHR_MIP_LDS"] = on
        r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **KW)
        for _ in range(30):
            r.render_async([6.0, 0.0, 0.5], 90.0, skip_bloom=True)
        ms = []
        for _ in range(20):
            r.render_async([6.0, 0.0, 0.5], 90.0, skip_bloom=True)
            ms.append(r.counters()["march_ms"])
        out[on] = (float(np.median(ms)), r.mip_lds_level())
        r.close()
    print(f"texture {tex_hw}: plain {out['0'][0]:.3f} ms; BHR_MIP_LDS=1 {out['1'][0]:.3f} ms (levels from {out['1'][1]} in LDS)", flush=True)
