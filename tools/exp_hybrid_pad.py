"""The strict band's tile padding (option hybrid_pad: share of a tile's own span of b it is padded by): per value, the fhd bench
frame's strict tiles, hybrid vs strict pixels (RMSE per channel, pixels beyond 1e-3 / 0.05), the isolated march time, and N random
views at fhd size (the generator of tests/test_gpu_fuzz.py) hybrid vs strict.
usage: python tools/exp_hybrid_pad.py [n_fuzz] [pads...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from bhr_amd import HipRenderer, _lib, scenes, workloads
from test_gpu_fuzz import _cases

n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pads = [float(x) for x in sys.argv[2:]] or [1.0, 0.5, 0.25]
wl = bench.WORKLOADS["fhd"]
r, _, _, _ = workloads.make_scene(wl, frame_slots=1)
r.render_async(wl["cam_pos"], wl["fov"], math="strict")
ref = {k: r.read_layer(v) for k, v in (("final", _lib.LAYER_FINAL), ("bg", _lib.LAYER_BG), ("disk", _lib.LAYER_DISK))}
for pad in pads:
    r.set_option("hybrid_pad", pad)
    for rep in (-1, 1):
        r.set_option("hybrid_repair", rep)
        r.render_async(wl["cam_pos"], wl["fov"], math="hybrid")
        got = {k: r.read_layer(v) for k, v in (("final", _lib.LAYER_FINAL), ("bg", _lib.LAYER_BG), ("disk", _lib.LAYER_DISK))}
        info = r.hybrid_info()
        d = np.abs(got["final"] - ref["final"]).max(axis=2)
        e = {k: [float(x) for x in np.sqrt(np.mean((got[k].astype(np.float64) - ref[k]) ** 2, axis=(0, 1)))] for k in got}
        print(json.dumps(dict(pad=pad, repair=rep, strict_tiles=info["strict_tiles"], beyond_1e3=int((d > 1e-3).sum()), beyond_5e2=int((d > 0.05).sum()), max=float(d.max()),
                              rmse_final=e["final"], rmse_disk=e["disk"])), flush=True)
    r.set_option("hybrid_repair", -1)
    for _ in range(20):
        r.render_async(wl["cam_pos"], wl["fov"], math="hybrid")
    r.timing_reset()
    for _ in range(60):
        r.render_async(wl["cam_pos"], wl["fov"], math="hybrid")
    c = r.counters()
    print(json.dumps(dict(pad=pad, march_ms=c["march_ms_sum"] / c["frames_timed"])), flush=True)
r.close()
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
for pad in pads:
    worst, flips, shares, bad = 0.0, 0, [], []
    for k, c in enumerate(_cases(n_fuzz, 23)):
        if c["kw"].get("anti_alias") == "lod_radius" and k % 2:
            continue
        q = HipRenderer(1920, 1080, sky, tex, math="hybrid", frame_slots=1, options={"hybrid_pad": pad}, **c["kw"])
        lay = {}
        for math in ("hybrid", "strict"):
            q.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True, math=math)
            lay[math] = (q.read_layer(_lib.LAYER_BG), q.read_layer(_lib.LAYER_DISK))
            if math == "hybrid":
                shares.append(q.hybrid_info()["strict_tiles"] / q.hybrid_info()["tiles"])
        q.close()
        e = max(float(np.sqrt(np.mean((lay["hybrid"][j].astype(np.float64) - lay["strict"][j]) ** 2, axis=(0, 1))).max()) for j in (0, 1))
        f = int(sum((np.abs(lay["hybrid"][j] - lay["strict"][j]).max(axis=2) > 0.05).sum() for j in (0, 1)))
        worst = max(worst, e); flips += f
        if e > 6e-5 or f:
            bad.append((k, e, f, float(np.linalg.norm(c["cam"]))))
    print(json.dumps(dict(pad=pad, fuzz_views=len(shares), worst_rmse=worst, flips=flips, mean_strict_share=float(np.mean(shares)), bad=bad[:8])), flush=True)
