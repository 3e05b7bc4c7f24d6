import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bhr_amd
from bhr_amd import HipRenderer, _lib, scenes, workloads
from test_reference_kernels import MARCH, KW, FLARE, load_scene, load_e2e, E2E_KW
def rm(a, b): return float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2, axis=(0, 1))).max())
for name in MARCH:
    g, sky, tex = load_scene(name)
    res = {}
    for math in ("strict", "fast"):
        hip = HipRenderer(int(g["width"]), int(g["height"]), sky, tex, lens_flare=(name in FLARE), math=math, **KW[name])
        out = hip.render(list(g["cam_pos"]), float(g["fov"]), frame=int(g["frame"]))
        res[math] = (rm(out, g["f32_final"]), rm(out, g["f64_final"]), float(np.abs(out - g["f32_final"]).max()), hip.counters()["ray_steps"])
        hip.close()
    print(f"{name:9s} strict vs f32 {res['strict'][0]:.2e} f64 {res['strict'][1]:.2e} | fast vs f32 {res['fast'][0]:.2e} f64 {res['fast'][1]:.2e} max {res['fast'][2]:.2e} steps {res['fast'][3]} / {int(g['f32_steps'].sum())}")
g, sky = load_e2e()
for math in ("strict", "fast"):
    hip = HipRenderer(320, 180, sky, g["disk_tex"], math=math, **E2E_KW)
    out = hip.render([6, 0, 0.5], 60); hip.close()
    print("e2e 320x180", math, f"rmse {rm(out, g['final']):.2e} max {np.abs(out - g['final']).max():.2e}")
# fhd bench scene: fast vs strict (strict == oracle to 5e-6, whole frame test)
wl = dict(width=1920, height=1080, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=0.0, anti_alias="disabled")
outs = {}
for math in ("strict", "fast"):
    hip, _, _, _ = workloads.make_scene(wl, math=math)
    outs[math] = hip.render(wl["cam_pos"], wl["fov"]); c = hip.counters(); outs[math + "_steps"] = c["ray_steps"]; hip.close()
d = np.abs(outs["fast"] - outs["strict"])
print(f"fhd bench frame fast vs strict: rmse {rm(outs['fast'], outs['strict'].astype(np.float64)):.2e} max {d.max():.2e} px>1e-3 {(d.max(axis=2) > 1e-3).sum()} steps {outs['fast_steps']} vs {outs['strict_steps']}")
for wl2, nm in ((dict(wl, width=3840, height=2160, disk_tilt=25.0, anti_alias="lod_radius"), "4k aa"),):
    o = {}
    for math in ("strict", "fast"):
        hip, _, _, _ = workloads.make_scene(wl2, math=math); o[math] = hip.render(wl2["cam_pos"], wl2["fov"]); hip.close()
    d = np.abs(o["fast"] - o["strict"]); print(f"{nm} fast vs strict: rmse {rm(o['fast'], o['strict'].astype(np.float64)):.2e} max {d.max():.2e}")
