#!/usr/bin/env python3
"""Where do hybrid and strict still differ on the 4k anti-aliased frame?  Worst pixels with their impact parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from bhr_amd import _lib, workloads
from bhr_amd.camera import build_camera

wl = bench.WORKLOADS["4k"]
r, _, _, _ = workloads.make_scene(wl, frame_slots=1)
lay = {}
for math in ("strict", "hybrid", "fast"):
    r.render_async(wl["cam_pos"], wl["fov"], math=math)
    lay[math] = dict(bg=r.read_layer(_lib.LAYER_BG), disk=r.read_layer(_lib.LAYER_DISK))
r.close()
W, H = wl["width"], wl["height"]
eye, right, up, fwd, pw, ph = build_camera(np.array(wl["cam_pos"], np.float64), wl["fov"], W, H)
for other in ("hybrid", "fast"):
    d = np.abs(lay[other]["disk"] - lay["strict"]["disk"]).max(axis=2)
    dbg = np.abs(lay[other]["bg"] - lay["strict"]["bg"]).max(axis=2)
    idx = np.argsort(d.ravel())[::-1][:40]
    print(other, "disk: n>1e-3", int((d > 1e-3).sum()), "bg: n>1e-3", int((dbg > 1e-3).sum()))
    for k in idx[:30]:
        y, x = divmod(int(k), W)
        tl = eye + fwd - (pw * W / 2) * right + (ph * H / 2) * up
        p = tl + (x + 0.5) * pw * right - (y + 0.5) * ph * up
        dv = (p - eye) / np.linalg.norm(p - eye)
        b = np.linalg.norm(np.cross(eye, dv))
        print(f"  ({x:4d},{y:4d}) b={b:.3f} d_disk={d[y, x]:.4f} strict={lay['strict']['disk'][y, x]} {other}={lay[other]['disk'][y, x]} bg_diff={dbg[y, x]:.2e}")
