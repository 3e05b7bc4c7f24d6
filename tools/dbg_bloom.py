"""Where the split post-pass differs from the exact one: tools/dbg_bloom.py [W H]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhr_amd import HipRenderer, scenes, _lib
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, outputs="f32+blur", **KW)
out = {}
r.set_option("bloom_split", 0)
r.render_async([6, 0, 0.5], 90)
out[0] = r.read_layer(_lib.LAYER_BLUR)
r.set_option("bloom_split", 1)
for dbg in (8, 128, 136, 4, 64, 0):
    r.set_option("bloom_dbg", dbg)
    r.render_async([6, 0, 0.5], 90)
    out[1] = r.read_layer(_lib.LAYER_BLUR)
    d = np.abs(out[1] - out[0])
    print("dbg", dbg, "max", d.max(), "count>3e-6", int((d > 3e-6).sum()))
for tiles in (1, 2, 4):
    r.set_option("bloom_dbg", 0)
    r.set_option("bloom_tiles", tiles)
    r.render_async([6, 0, 0.5], 90)
    out[1] = r.read_layer(_lib.LAYER_BLUR)
    d = np.abs(out[1] - out[0])
    print("tiles", tiles, "max", d.max(), "count>3e-6", int((d > 3e-6).sum()))
ys, xs, cs = np.nonzero(d > 3e-6)
for y, x, c in list(zip(ys, xs, cs))[:40]:
    print(y, x, c, out[1][y, x, c], out[0][y, x, c])
if len(ys):
    print("rows", np.unique(ys)[:50], "cols", np.unique(xs)[:50], "ch", np.unique(cs))
