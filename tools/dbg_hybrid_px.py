"""Where hybrid and strict differ most on the fhd bench frame: pixel, impact parameter of its ray, layer.
usage: python tools/dbg_hybrid_px.py [band_hi]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bhr_amd import _lib, workloads

wl = bench.WORKLOADS["fhd"]
r, _, _, _ = workloads.make_scene(wl, frame_slots=1)
for a in sys.argv[1:]:
    if a.startswith("repair="):
        r.set_option("hybrid_repair", int(a.split("=")[1]))
    else:
        r.set_option("hybrid_band_hi", float(a))
lay = {}
for math in ("strict", "hybrid"):
    r.render_async(wl["cam_pos"], wl["fov"], math=math)
    lay[math] = {k: r.read_layer(v) for k, v in (("final", _lib.LAYER_FINAL), ("bg", _lib.LAYER_BG), ("disk", _lib.LAYER_DISK))}
print(r.hybrid_info())
cam = r.camera_uniforms(wl["cam_pos"], wl["fov"])
W, H = wl["width"], wl["height"]
cp = np.array(list(cam.pos), np.float64); cf = np.array(list(cam.forward)); cr = np.array(list(cam.right)); cu = np.array(list(cam.up))
tl = cp + cf - cam.pixel_width * W / 2 * cr + cam.pixel_height * H / 2 * cu
for k in ("final", "bg", "disk"):
    d = np.abs(lay["hybrid"][k] - lay["strict"][k]).max(axis=2)
    idx = np.argsort(d.ravel())[::-1][:8]
    print(k, "count>1e-3", int((d > 1e-3).sum()), "count>3e-4", int((d > 3e-4).sum()))
    for q in idx:
        y, x = divmod(int(q), W)
        dv = tl + (x + 0.5) * cam.pixel_width * cr - (y + 0.5) * cam.pixel_height * cu - cp
        dv /= np.linalg.norm(dv)
        pd = cp @ dv
        bl2 = cp @ cp - pd * pd
        b = 1 / np.sqrt(1 / bl2 - 1 / (cp @ cp) ** 1.5)
        print(f"   ({x},{y}) |d|={d[y, x]:.3g} b-b_c={b - 2.598076211353316:+.4f} strict={lay['strict'][k][y, x]} hybrid={lay['hybrid'][k][y, x]}")
r.close()
