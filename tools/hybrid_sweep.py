#!/usr/bin/env python3
"""Band sweep for math="hybrid" (GPU): for every band [b_c - lo, b_c + hi] of impact parameters that is marched
strict, the distance of the hybrid frame from the strict frame (= the reference's f32 statements to 3e-7) and from the
reference-statement fixtures, the ray-step totals, the share of strict tiles and the march time.

usage: python tools/hybrid_sweep.py [--out gpurun_out/hybrid_sweep.json] [--bands "0,0;0.05,0.1;..."] [--scenes fhd,e2e,fixtures,4k]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def rmse_c(a, b):
    return np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2, axis=(0, 1)))


def stats(img, ref):
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    return {"rmse": [float(x) for x in rmse_c(img, ref)], "max": float(d.max()), "n_gt_1e-3": int((d.max(axis=2) > 1e-3).sum()),
            "n_gt_1e-4": int((d.max(axis=2) > 1e-4).sum())}


def timed(r, cam, fov, math, n=30, **kw):
    for _ in range(5):
        r.render_async(cam, fov, math=math, **kw)
    r.timing_reset()
    r.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_async(cam, fov, math=math, **kw)
    r.sync()
    wall = (time.perf_counter() - t0) / n * 1e3
    c = r.counters()
    return {"march_ms": c["march_ms_sum"] / max(c["frames_timed"], 1), "post_ms": c["bloom_ms_sum"] / max(c["frames_timed"], 1),
            "wall_ms": wall, "ray_steps": int(c["ray_steps"])}


def run_scene(name, make, bands, layers=True, n_time=30):
    """make() -> (renderer factory taking no args, cam, fov, kwargs for render, dict of reference frames)"""
    from bhr_amd import _lib
    out = {"scene": name, "bands": []}
    os.environ.pop("BHR_HYBRID_BAND", None)
    r, cam, fov, kw, refs = make()
    r.render_async(cam, fov, math="strict", **kw)
    strict = {"final": r.read_layer(_lib.LAYER_FINAL), "disk": r.read_layer(_lib.LAYER_DISK), "bg": r.read_layer(_lib.LAYER_BG)}
    out["strict"] = timed(r, cam, fov, "strict", n_time, **kw)
    r.render_async(cam, fov, math="fast", **kw)
    fast = {"final": r.read_layer(_lib.LAYER_FINAL), "disk": r.read_layer(_lib.LAYER_DISK), "bg": r.read_layer(_lib.LAYER_BG)}
    out["fast"] = dict(timed(r, cam, fov, "fast", n_time, **kw), vs_strict={k: stats(fast[k], strict[k]) for k in strict})
    for k, ref in refs.items():
        out.setdefault("strict_vs_ref", {})[k] = stats(strict["final"], ref)
        out.setdefault("fast_vs_ref", {})[k] = stats(fast["final"], ref)
    r.close()
    for lo, hi in bands:
        os.environ["BHR_HYBRID_BAND"] = f"{lo},{hi}"
        r, cam, fov, kw, refs = make()
        r.render_async(cam, fov, math="hybrid", **kw)
        hy = {"final": r.read_layer(_lib.LAYER_FINAL), "disk": r.read_layer(_lib.LAYER_DISK), "bg": r.read_layer(_lib.LAYER_BG)}
        info = r.hybrid_info()
        t = timed(r, cam, fov, "hybrid", n_time, **kw)
        row = {"lo": lo, "hi": hi, "strict_tile_share": info["strict_tiles"] / info["tiles"], **t,
               "steps_rel": abs(t["ray_steps"] - out["strict"]["ray_steps"]) / out["strict"]["ray_steps"],
               "vs_strict": {k: stats(hy[k], strict[k]) for k in strict}}
        for k, ref in refs.items():
            row.setdefault("vs_ref", {})[k] = stats(hy["final"], ref)
        out["bands"].append(row)
        r.close()
        print(f"[{name}] band -{lo}/+{hi}: strict tiles {row['strict_tile_share']:.3f} march {t['march_ms']:.3f} ms "
              f"rmse(final) {max(row['vs_strict']['final']['rmse']):.3g} max {row['vs_strict']['final']['max']:.3g} "
              f">1e-3: {row['vs_strict']['final']['n_gt_1e-3']} steps_rel {row['steps_rel']:.2e}", flush=True)
    os.environ.pop("BHR_HYBRID_BAND", None)
    print(f"[{name}] strict {out['strict']['march_ms']:.3f} ms, fast {out['fast']['march_ms']:.3f} ms, fast vs strict rmse "
          f"{max(out['fast']['vs_strict']['final']['rmse']):.3g}", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hybrid_sweep.json"))
    ap.add_argument("--bands", default="0,0;0.03,0.05;0.06,0.1;0.1,0.2;0.12,0.3;0.2,0.4;0.3,0.6;0.5,1.0")
    ap.add_argument("--scenes", default="fhd,e2e,fixtures,4k")
    a = ap.parse_args()
    bands = [tuple(float(x) for x in b.split(",")) for b in a.bands.split(";")]
    scenes = a.scenes.split(",")
    import bench
    from bhr_amd import HipRenderer, workloads
    res = []

    if "fhd" in scenes:
        def mk():
            wl = bench.WORKLOADS["fhd"]
            r, _, _, _ = workloads.make_scene(wl, frame_slots=1)
            return r, wl["cam_pos"], wl["fov"], {}, {}
        res.append(run_scene("fhd", mk, bands))
    if "e2e" in scenes:
        from test_reference_kernels import E2E_KW, load_e2e
        g, sky = load_e2e()

        def mk():
            return HipRenderer(320, 180, sky, g["disk_tex"], frame_slots=1, **E2E_KW), [6, 0, 0.5], 60, {}, {"f32": g["final"]}
        res.append(run_scene("e2e", mk, bands, n_time=10))
    if "fixtures" in scenes:
        from test_reference_kernels import FLARE, KW, MARCH, load_scene
        for name in MARCH:
            g, sky, tex = load_scene(name)

            def mk(name=name, g=g, sky=sky, tex=tex):
                r = HipRenderer(int(g["width"]), int(g["height"]), sky, tex, lens_flare=(name in FLARE), frame_slots=1, **KW[name])
                return r, list(g["cam_pos"]), float(g["fov"]), {"frame": int(g["frame"])}, {"f32": g["f32_final"], "f64": g["f64_final"]}
            res.append(run_scene("fixture_" + name, mk, bands, n_time=5))
    if "4k" in scenes:
        def mk():
            wl = bench.WORKLOADS["4k"]
            r, _, _, _ = workloads.make_scene(wl, frame_slots=1)
            return r, wl["cam_pos"], wl["fov"], {"lens_flare": False}, {}
        res.append(run_scene("4k_tilt_aa", mk, bands, n_time=10))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
