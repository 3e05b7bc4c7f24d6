#!/usr/bin/env python3
"""Distance of each march arithmetic from the reference-statement fixtures, per view and layer (GPU): the table behind
the bars of tests/test_gpu_reference_kernels.py::test_kernels_vs_reference_statements_f64."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_reference_kernels import FLARE, KW, MARCH, load_scene
from bhr_amd import HipRenderer, _lib


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2, axis=(0, 1))).max())


def ref(g, mode, k):
    a = g[f"{mode}_{k}"]
    return a if k == "final" else a.transpose(1, 0, 2)


out = {}
for name in MARCH:
    g, sky, tex = load_scene(name)
    for math in ("strict", "hybrid", "fast"):
        hip = HipRenderer(int(g["width"]), int(g["height"]), sky, tex, lens_flare=(name in FLARE), math=math, **KW[name])
        final = hip.render(list(g["cam_pos"]), float(g["fov"]), frame=int(g["frame"]))
        lay = dict(final=final, bg=hip.read_layer(_lib.LAYER_BG), disk=hip.read_layer(_lib.LAYER_DISK), blur=hip.read_layer(_lib.LAYER_BLUR))
        hip.close()
        for k in ("bg", "disk", "blur", "final"):
            out[f"{name}/{math}/{k}"] = {"vs_f64": rmse(lay[k], ref(g, "f64", k)), "vs_f32": rmse(lay[k], ref(g, "f32", k)),
                                         "f32_vs_f64": rmse(ref(g, "f32", k), ref(g, "f64", k))}
            print(f"{name:10s} {math:7s} {k:6s} vs f64 {out[f'{name}/{math}/{k}']['vs_f64']:.3g}  vs f32 {out[f'{name}/{math}/{k}']['vs_f32']:.3g}  "
                  f"(reference f32 vs f64: {out[f'{name}/{math}/{k}']['f32_vs_f64']:.3g})", flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fast_bars.json"), "w"), indent=1)
