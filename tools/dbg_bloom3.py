"""Decode the split post-pass's packed intermediates and compare them with numpy: which stage loses bits?"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhr_amd import HipRenderer, scenes, _lib
W, H = 1920, 1080
KW = dict(step_size=0.5, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
r = HipRenderer(W, H, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), math="fast", frame_slots=1, **KW)
r.render_async([6, 0, 0.5], 90, skip_bloom=True)
r.write_layer(_lib.LAYER_BG, np.zeros((H, W, 3), np.float32))
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
x = np.stack([0.5 + 0.5 * np.sin(xx / 50) * np.cos(yy / 70), 0.3 + 0.2 * np.cos(xx / 33), 0.6 + 0.4 * np.sin(yy / 21)], axis=2).astype(np.float32)
r.set_option("bloom_split", 1)
r.write_layer(_lib.LAYER_DISK, x)
r.bloom_only()
geom = (C.c_int32 * 10)()
lib = _lib.load()
_lib.check(lib.bhr_debug_read(r._ctx, 0, None, 0, geom))
NT, n_tx, WP, YB, GP, g0, t_first, n_ty, pbr, GR = list(geom)
print("geom", list(geom))
pa = np.empty(6 * YB * GP * 256, np.float16)
_lib.check(lib.bhr_debug_read(r._ctx, 0, pa.ctypes.data, pa.nbytes, None))
pb = np.empty(6 * GR * WP * 8, np.float16)
_lib.check(lib.bhr_debug_read(r._ctx, 1, pb.ctypes.data, pb.nbytes, None))
pa = pa.reshape(3, 2, YB, GP, 32, 8).astype(np.float64)
v = (pa[:, 0] + pa[:, 1]) / 16384.0                       # [c][yb][g][yy][j]
v = v[:, :, g0:g0 + W // 8].transpose(1, 3, 2, 4, 0).reshape(YB * 32, W, 3)[:H]
print("pa vs input: max", np.abs(v - x).max())
pb = pb.reshape(3, 2, WP // 32, GR, 32, 8).astype(np.float64)
hb = (pb[:, 0] + pb[:, 1]) / 16384.0                      # [c][strip][gr][n][j]
hb = hb.transpose(2, 4, 1, 3, 0).reshape(GR * 8, WP, 3)
r0 = 0 - pbr
hb = hb[r0:r0 + H, :W]
# numpy H pass
R = int(W * 0.02); s = np.float32((W / 640.0) ** 2)
acc = np.zeros((H, W, 3)); ws = np.zeros((W, 3))
xd = x.astype(np.float64)
for d in range(-R, R + 1):
    w = np.array([np.float64(np.exp(np.float32(-np.float32(d * d) / (np.float32(sg) * s)))) for sg in (25.0, 80.0, 1600.0)])
    lo, hi = max(0, -d), min(W, W - d)
    acc[:, lo:hi] += xd[:, lo + d:hi + d] * w
    ws[lo:hi] += w
ref = acc / ws[None]
dd = np.abs(hb - ref)
print("pb vs numpy H pass: max", dd.max(), "count>1e-6", int((dd > 1e-6).sum()))
ys, xs, cs = np.nonzero(dd > 1e-6)
for y, xx_, c in list(zip(ys, xs, cs))[:12]:
    h_, l_ = pb[c, 0, xx_ // 32, (y + r0) // 8, xx_ % 32, (y + r0) % 8], pb[c, 1, xx_ // 32, (y + r0) // 8, xx_ % 32, (y + r0) % 8]
    print(y, xx_, c, "got", hb[y, xx_, c], "ref", ref[y, xx_, c], "hi", h_, "lo", l_, "ref*16384", ref[y, xx_, c] * 16384)
