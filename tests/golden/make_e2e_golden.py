#!/usr/bin/env python3
"""Generates tests/golden/e2e_ref.npz: the reference's tests/e2e_render.py frame (320x180, configs[0] of
BASELINE.json) produced by the reference's OWN `render_image` -- procedural skybox, entity lifecycle,
background / compose / mip kernels, ray march, bloom -- with its Taichi kernels running as plain Python on
tests/golden/ti_shim.py in binary32 mode (see make_kernel_golden.py).  ~15 minutes on one core.

Run in the build container only:  python tests/golden/make_e2e_golden.py
Stored: the (180, 320, 3) f32 frame, the disk texture the lifecycle produced, the statistics, hashes of the
inputs, and the MD5 of the frame bytes next to the reference's tests/e2e_baseline.txt (an MD5 of the author's
Taichi/LLVM fast-math build: not expected to match any other evaluation order).
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import ti_shim  # noqa: E402

ti_shim.install()
ti_shim.set_default_fp("f32")
sys.path.insert(0, REF)
import render as ref  # noqa: E402

# tests/e2e_render.py:27-43
PARAMS = dict(width=320, height=180, cam_pos=[6, 0, 0.5], fov=60, step_size=0.1, r_max=10, device="cpu",
              n_stars=100, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15, lens_flare=False,
              anti_alias="disabled", force_regenerate_disk_texture=True, ignore_taichi_cache=True)


def main():
    out_dir = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE
    keep = {}
    orig_init = ref.TaichiRenderer.__init__

    def spy_init(self, *a, **k):
        orig_init(self, *a, **k)
        keep["renderer"] = self
        keep["sky_sha256"] = hashlib.sha256(np.ascontiguousarray(a[2]).tobytes()).hexdigest()

    ref.TaichiRenderer.__init__ = spy_init
    t0 = time.time()
    img = ref.render_image(**PARAMS)
    r = keep["renderer"]
    print(f"render_image: {time.time() - t0:.0f} s", flush=True)
    img = np.ascontiguousarray(img, dtype=np.float32)
    md5 = hashlib.md5(img.tobytes()).hexdigest()
    with open(os.path.join(REF, "tests", "e2e_baseline.txt")) as f:
        baseline = f.read().strip()
    print("md5 of this evaluation:", md5, " reference baseline:", baseline)
    np.savez_compressed(os.path.join(out_dir, "e2e_ref.npz"), final=img, md5=md5, baseline_md5=baseline,
                        sky_sha256=keep["sky_sha256"], disk_tex=r.disk_texture_field.to_numpy(),
                        stats=r._param_stats_field.to_numpy(), row_stats=r._param_row_stats_field.to_numpy())


if __name__ == "__main__":
    main()
