"""Randomised check of the device PNG encoder: frames of random sizes (odd widths: the unaligned row loader) and
contents (noise, gradients, sparse stars, saturated) -> every file parsed (all CRCs, Adler-32) and decoded with PIL.
Usage: python tools/fuzz_png.py [n_cases] [seed]"""
import io, os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from bhr_amd import HipRenderer, _lib, scenes
from bhr_amd.output import png_encode_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
sky, tex = scenes.analytic_skybox(32, 64), scenes.noisy_disk()
bad = 0
for k in range(n):
    w, h = int(rng.integers(1, 900)), int(rng.integers(1, 48))
    kind = k % 5
    if kind == 0: f = rng.random((h, w, 3), dtype=np.float32)
    elif kind == 1: f = np.broadcast_to(np.linspace(0, 1, w, dtype=np.float32)[None, :, None] * rng.random(3, dtype=np.float32), (h, w, 3)).copy()
    elif kind == 2: f = (rng.random((h, w, 3)) > 0.98).astype(np.float32) * rng.random((h, w, 3), dtype=np.float32)
    elif kind == 3: f = np.clip(rng.normal(0.5, 0.02, (h, w, 3)), 0, 1).astype(np.float32)
    else: f = np.full((h, w, 3), float(rng.integers(0, 2)), np.float32)
    r = HipRenderer(w, h, sky, tex)
    r.render_async([6, 0, 0.5], 90)
    r.write_layer(_lib.LAYER_FINAL, f)
    data = png_encode_device(r)
    want = (np.clip(f, 0, 1) * 255).astype(np.uint8)
    ok = data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat = 8, []
    while at < len(data) and ok:
        ln, typ = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + ln]
        ok = ok and zlib.crc32(typ + body) == struct.unpack(">I", data[at + 8 + ln:at + 12 + ln])[0]
        if typ == b"IDAT": idat.append(body)
        at += 12 + ln
    ok = ok and len(zlib.decompress(b"".join(idat))) == h * (3 * w + 1)
    ok = ok and np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want)
    ok = ok and len(data) <= _lib.load().bhr_png_device_bound(w, h)
    if not ok:
        bad += 1
        print("FAILED", k, w, h, kind, flush=True)
    r.close()
print(f"{n} random frames: {bad} failures")
