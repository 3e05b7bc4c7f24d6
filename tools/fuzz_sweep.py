"""Large randomised parity sweep (not a test: a one-off run recorded in DESIGN 2): N random views, as
tests/test_gpu_fuzz.py draws them, strict HIP march against the oracle -- equal ray-step totals, layer differences.
Usage: python tools/fuzz_sweep.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from bhr_amd import HipRenderer, _lib, scenes
from test_gpu_fuzz import _cases
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
w, h = 64, 40
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
bad, worst, steps_total = [], 0.0, 0
for k, c in enumerate(_cases(n, seed)):
    hip = HipRenderer(w, h, sky, tex, **c["kw"])
    ora = oracle.OracleRenderer(w, h, sky, tex, **c["kw"])
    hip.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True)
    bg, disk = hip.read_layer(_lib.LAYER_BG), hip.read_layer(_lib.LAYER_DISK)
    rbg, rdisk = (x.transpose(1, 0, 2) for x in ora.march(c["cam"], c["fov"], frame=c["frame"]))
    ok_steps = hip.counters()["ray_steps"] == ora.last_total_steps
    rm = max(float(np.sqrt(np.mean((disk - rdisk) ** 2))), float(np.sqrt(np.mean((bg - rbg) ** 2))))
    mx = max(float(np.abs(bg - rbg).max()), float(np.abs(disk - rdisk).max()))
    worst = max(worst, rm)
    steps_total += ora.last_total_steps
    if not ok_steps or rm > 1e-5 or mx > 2e-4 or not np.isfinite(bg).all():
        bad.append((k, ok_steps, rm, mx, c))
    hip.close()
    if (k + 1) % 100 == 0:
        print(f"{k + 1} views, {len(bad)} outside the bounds, worst RMSE {worst:.2e}", flush=True)
print(f"{n} random views (seed {seed}, {w}x{h}, {steps_total / 1e6:.1f} M ray-steps): {len(bad)} outside the bounds "
      f"(equal step totals, RMSE <= 1e-5, max <= 2e-4); worst RMSE {worst:.2e}")
for b in bad[:10]:
    print(b)
