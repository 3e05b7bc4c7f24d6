import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhr_amd import HipRenderer, scenes, _lib
W, H = 1920, 1080
KW = dict(step_size=0.5, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
r = HipRenderer(W, H, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), math="fast", frame_slots=1, **KW)
r.render_async([6, 0, 0.5], 90, skip_bloom=True)
r.write_layer(_lib.LAYER_BG, np.zeros((H, W, 3), np.float32))
def bloom(x, split):
    r.set_option("bloom_split", split)
    r.write_layer(_lib.LAYER_DISK, x)
    r.bloom_only()
    return r.read_layer(_lib.LAYER_BLUR)
rng = np.random.default_rng(1)
for name, x in (("random", rng.random((H, W, 3), dtype=np.float32)),
                ("one pixel", None), ("smooth", None)):
    if name == "one pixel":
        x = np.zeros((H, W, 3), np.float32); x[500, 1000] = 1.0; x[200, 300] = 0.37
    if name == "smooth":
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        x = np.stack([0.5 + 0.5 * np.sin(xx / 50) * np.cos(yy / 70), 0.3 + 0.2 * np.cos(xx / 33), 0.6 + 0.4 * np.sin(yy / 21)], axis=2).astype(np.float32)
    a, b = bloom(x, 1), bloom(x, 0)
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), 1e-12)
    ys, xs, cs = np.nonzero(d > 3e-6)
    print(name, "max abs", d.max(), "max rel", rel[b > 1e-6].max() if (b > 1e-6).any() else 0, "count", len(ys), "rows", np.unique(ys)[:12], len(np.unique(ys)), "cols", np.unique(xs)[:12], len(np.unique(xs)))

def np_bloom(x):
    R = int(W * 0.02); s = (W / 640.0) ** 2
    x = x.astype(np.float64)
    out = x
    for axis, n in ((1, W), (0, H)):
        acc = np.zeros_like(out); ws = np.zeros((n, 3))
        for d in range(-R, R + 1):
            w = np.array([np.exp(-np.float32(d * d) / np.float32(sg * np.float32(s))) for sg in (25.0, 80.0, 1600.0)], dtype=np.float64)
            w = np.array([np.float64(np.float32(np.exp(np.float32(-np.float32(d * d) / (np.float32(sg) * np.float32(s)))))) for sg in (25.0, 80.0, 1600.0)])
            lo, hi = max(0, -d), min(n, n - d)
            if axis == 1:
                acc[:, lo:hi] += out[:, lo + d:hi + d] * w
            else:
                acc[lo:hi] += out[lo + d:hi + d] * w
            ws[lo:hi] += w
        out = acc / (ws[None, :, :] if axis == 1 else ws[:, None, :])
    return out
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
x = np.stack([0.5 + 0.5 * np.sin(xx / 50) * np.cos(yy / 70), 0.3 + 0.2 * np.cos(xx / 33), 0.6 + 0.4 * np.sin(yy / 21)], axis=2).astype(np.float32)
ref = np_bloom(x)
for split in (1, 0):
    got = bloom(x, split)
    d = np.abs(got - ref)
    ys, xs, cs = np.nonzero(d > 3e-6)
    print("vs numpy f64: split", split, "max", d.max(), "count", len(ys), "cols", np.unique(xs)[:20], len(np.unique(xs)), "ch", np.unique(cs))
