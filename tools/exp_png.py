"""Device PNG encoder at the BASELINE frame sizes: time per frame (synchronous call: quantise + 3 launches + fetch of
the file), file size against zlib level 1 / 6 on the host.  Usage: python tools/exp_png.py [fhd|4k|8k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhr_amd import workloads
from bhr_amd.output import png_encode, png_encode_device
WL = {"fhd": dict(width=1920, height=1080, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=0.0, anti_alias="disabled"),
      "4k": dict(width=3840, height=2160, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=25.0, anti_alias="lod_radius"),
      "8k": dict(width=7680, height=4320, cam_pos=[6, 0, 0.5], fov=90, step_size=0.05, disk_tilt=0.0, anti_alias="disabled")}
for name in (sys.argv[1:] or ["fhd", "4k"]):
    wl = WL[name]
    r, _, _, _ = workloads.make_scene(wl)
    r.render_async(wl["cam_pos"], wl["fov"])
    u8 = r.read_final_u8()
    d = png_encode_device(r)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        d = png_encode_device(r)
    dt = (time.perf_counter() - t0) / n
    from PIL import Image
    import io
    ok = np.array_equal(np.asarray(Image.open(io.BytesIO(d)).convert("RGB")), u8)
    t0 = time.perf_counter(); z1 = png_encode(u8, 1, 1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); z6 = png_encode(u8, 6, 1); t6 = time.perf_counter() - t0
    print(f"{name}: device {dt * 1e3:.3f} ms, {len(d) / 1e6:.3f} MB (decodes to the frame: {ok}); zlib level 1 {t1 * 1e3:.1f} ms "
          f"{len(z1) / 1e6:.3f} MB; level 6 {t6 * 1e3:.1f} ms {len(z6) / 1e6:.3f} MB; raw {u8.nbytes / 1e6:.2f} MB", flush=True)
    r.close()
