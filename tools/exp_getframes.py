import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bhr_amd import workloads
for name, wl in (("fhd", dict(width=1920, height=1080, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=0.0, anti_alias="disabled")),
                 ("4k", dict(width=3840, height=2160, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=25.0, anti_alias="lod_radius"))):
    r, _, _, _ = workloads.make_scene(wl)
    r.render_async(wl["cam_pos"], wl["fov"])
    u8 = r.read_final_u8()
    np.save(f"gpurun_out/frame_{name}.npy", u8)
    print(name, u8.shape, u8.mean())
    r.close()
