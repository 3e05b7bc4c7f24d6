"""One view of the hybrid fuzz sweep again, under several values of one option: python tools/dbg_fuzz_view.py <n> <seed> <k> <W> <H> <option> <v0> <v1> ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bhr_amd import HipRenderer, _lib, scenes
from test_gpu_fuzz import _cases
n, seed, k, w, h = (int(x) for x in sys.argv[1:6])
opt, vals = sys.argv[6], [float(x) for x in sys.argv[7:]]
c = list(_cases(n, seed))[k]
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
for v in vals:
    r = HipRenderer(w, h, sky, tex, math="hybrid", options={opt: v}, **c["kw"])
    lay = {}
    for math in ("hybrid", "strict", "fast"):
        r.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True, math=math)
        lay[math] = (r.read_layer(_lib.LAYER_BG), r.read_layer(_lib.LAYER_DISK))
        if math == "hybrid":
            info = r.hybrid_info()
    r.close()
    e = {m: max(float(np.sqrt(np.mean((lay[m][j].astype(np.float64) - lay["strict"][j]) ** 2, axis=(0, 1))).max()) for j in (0, 1)) for m in ("hybrid", "fast")}
    print(json.dumps({opt: v, "rmse_hybrid_vs_strict": e["hybrid"], "rmse_fast_vs_strict": e["fast"], "strict_tiles": info["strict_tiles"], "tiles": info["tiles"]}), flush=True)
