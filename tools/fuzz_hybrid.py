"""Large randomised sweep of math="hybrid" against the strict march of the same context (not a test: a one-off run
recorded in DESIGN 2): N random views as tests/test_gpu_fuzz.py draws them -- cameras from 1.15 to 60 r_s, any tilt,
three step sizes, both AA modes -- at 192x128 (384 tiles).  Per view: per-channel RMSE of both layers, pixels that differ
by more than 0.05 (a ray that hits the disk under one arithmetic and not the other), ray-step totals, share of strict tiles.
Usage: python tools/fuzz_hybrid.py [n_cases] [seed] [--out file.json] [--size W H]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bhr_amd import HipRenderer, _lib, scenes
from test_gpu_fuzz import _cases

_skip = set()
for _f, _n in (("--out", 1), ("--size", 2)):
    if _f in sys.argv:
        _i = sys.argv.index(_f)
        _skip.update(range(_i, _i + _n + 1))
args = [a for k, a in enumerate(sys.argv) if k > 0 and k not in _skip]
n = int(args[0]) if len(args) > 0 else 1000
seed = int(args[1]) if len(args) > 1 else 11
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "fuzz_hybrid.json")
w, h = (int(sys.argv[sys.argv.index("--size") + 1]), int(sys.argv[sys.argv.index("--size") + 2])) if "--size" in sys.argv else (192, 128)
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
rows, bad = [], []
for k, c in enumerate(_cases(n, seed)):
    r = HipRenderer(w, h, sky, tex, math="hybrid", **c["kw"])
    lay, steps = {}, {}
    for math in ("hybrid", "strict"):
        r.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True, math=math)
        lay[math] = (r.read_layer(_lib.LAYER_BG), r.read_layer(_lib.LAYER_DISK))
        steps[math] = r.counters()["ray_steps"]
        if math == "hybrid":
            info = r.hybrid_info()
    r.close()
    e = max(float(np.sqrt(np.mean((lay["hybrid"][j].astype(np.float64) - lay["strict"][j]) ** 2, axis=(0, 1))).max()) for j in (0, 1))
    flips = int(sum((np.abs(lay["hybrid"][j] - lay["strict"][j]).max(axis=2) > 0.05).sum() for j in (0, 1)))
    dstep = abs(steps["hybrid"] - steps["strict"]) / max(steps["strict"], 1)
    row = dict(k=k, r_cam=float(np.linalg.norm(c["cam"])), rmse=e, flips=flips, step_rel=dstep,
               strict_share=info["strict_tiles"] / max(info["tiles"], 1), repaired_share=info["repaired_pixels"] / (w * h),
               repair_overflow=bool(info["repaired_pixels"] > info["repair_capacity"] > 0), finite=bool(np.isfinite(lay["hybrid"][0]).all() and np.isfinite(lay["hybrid"][1]).all()))
    rows.append(row)
    if e > 6e-5 or flips or dstep > 2e-4 or not row["finite"]:
        bad.append(dict(row, case={kk: (vv if not isinstance(vv, dict) else {a: (float(b) if not isinstance(b, str) else b) for a, b in vv.items()}) for kk, vv in c.items() if kk != "cam"},
                        cam=[float(x) for x in c["cam"]]))
    if (k + 1) % 100 == 0:
        print(f"{k + 1} views: {len(bad)} outside the bounds, worst RMSE {max(x['rmse'] for x in rows):.2e}", flush=True)
rm = np.array([x["rmse"] for x in rows])
summary = dict(n=n, seed=seed, size=[w, h], outside=len(bad), worst_rmse=float(rm.max()), median_rmse=float(np.median(rm)),
               q99_rmse=float(np.quantile(rm, 0.99)), over_3e5=int((rm > 3e-5).sum()), flips=int(sum(x["flips"] for x in rows)),
               mean_strict_share=float(np.mean([x["strict_share"] for x in rows])), max_repaired_share=float(max(x["repaired_share"] for x in rows)),
               repair_overflows=int(sum(x["repair_overflow"] for x in rows)), bounds="RMSE <= 6e-5, no pixel beyond 0.05, step totals within 2e-4")
print(json.dumps(summary))
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(dict(summary=summary, bad=bad[:50]), open(out, "w"), indent=1)
