#!/usr/bin/env python3
"""One fuzzed view (tests/test_gpu_fuzz.py::_cases(n, seed)[k]) under strict / hybrid / fast and hybrid with a wide band:
disk-layer RMSE and counts of differing pixels against strict, and where the hybrid differences sit in impact parameter.
usage: python tools/dbg_hybrid_view.py k n seed [width height]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bhr_amd import HipRenderer, _lib, scenes
from bhr_amd.camera import build_camera
from test_gpu_fuzz import _cases
k, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (192, 128)
c = _cases(n, seed)[k]
for kv in sys.argv[6:]:                     # overrides, e.g. anti_alias=disabled step_size=0.1
    kk, vv = kv.split("=")
    c["kw"][kk] = vv if kk == "anti_alias" else float(vv)
print(c)
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()


def layers(math, env=None):
    for kk, v in (env or {}).items():
        os.environ[kk] = v
    r = HipRenderer(w, h, sky, tex, math=math, **c["kw"])
    r.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True)
    out = r.read_layer(_lib.LAYER_DISK), r.read_layer(_lib.LAYER_BG), (r.hybrid_info() if math == "hybrid" else None)
    r.close()
    for kk in (env or {}):
        os.environ.pop(kk)
    return out


ref = layers("strict")
rm = lambda a, b: float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2, axis=(0, 1))).max())
for tag, math, env in (("hybrid", "hybrid", None), ("fast", "fast", None), ("hybrid band 0.6,1.2", "hybrid", {"BHR_HYBRID_BAND": "0.6,1.2"}),
                       ("hybrid no repair", "hybrid", {"BHR_HYBRID_REPAIR": "0"}), ("hybrid band 5,50 (all strict)", "hybrid", {"BHR_HYBRID_BAND": "5,50"})):
    d, b, info = layers(math, env)
    dd = np.abs(d - ref[0]).max(axis=2)
    print(f"{tag:32s} disk RMSE {rm(d, ref[0]):.3g} bg RMSE {rm(b, ref[1]):.3g}  px>1e-4 {int((dd > 1e-4).sum())} >1e-3 {int((dd > 1e-3).sum())} >1e-2 {int((dd > 1e-2).sum())}  {info}")
    if tag == "hybrid":
        eye, right, up, fwd, pw, ph = build_camera(np.array(c["cam"], np.float64), c["fov"], w, h)
        r0 = np.linalg.norm(eye)
        idx = np.argsort(dd.ravel())[::-1][:12]
        for q in idx:
            y, x = divmod(int(q), w)
            tl = eye + fwd - (pw * w / 2) * right + (ph * h / 2) * up
            p = tl + (x + 0.5) * pw * right - (y + 0.5) * ph * up
            dv = (p - eye) / np.linalg.norm(p - eye)
            bl = np.linalg.norm(np.cross(eye, dv))
            be = 1 / np.sqrt(max(1 / bl ** 2 - 1 / r0 ** 3, 1e-9))
            # straight-line hit on the tilted disk plane (bending neglected: these rays pass far from the hole)
            tl_ = np.radians(c["kw"]["disk_tilt"])
            nrm = np.array([0.0, -np.sin(tl_), np.cos(tl_)])               # plane z cos(t) - y sin(t) = 0
            tt = -np.dot(eye, nrm) / np.dot(dv, nrm)
            hp = eye + tt * dv
            yd = hp[1] * np.cos(tl_) + hp[2] * np.sin(tl_)                 # in-plane coordinates
            print(f"   ({x:3d},{y:3d}) b_l {bl:.3f} b {be:.3f} diff {dd[y, x]:.4f} flat hit r {np.hypot(hp[0], yd):.2f} phi {np.degrees(np.arctan2(yd, hp[0])):.1f} deg t {tt:.1f}  strict {ref[0][y, x]} hybrid {d[y, x]}")
