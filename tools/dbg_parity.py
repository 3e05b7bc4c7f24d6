import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd
from bhr_amd import HipRenderer, scenes, _lib
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "default"
edge = not (len(sys.argv) > 2 and sys.argv[2] == "hard")
s = scenes.SCENES[name]
sky, tex = scenes.analytic_skybox(), scenes.noisy_disk(edge=edge)
ora = O.OracleRenderer(s["width"], s["height"], sky, tex, **s["kw"])
ref, rbg, rdisk, rblur = ora.render(s["cam_pos"], s["fov"], parts=True)
rbg, rdisk, rblur = (x.transpose(1, 0, 2) for x in (rbg, rdisk, rblur))
truth = O.OracleRenderer(s["width"], s["height"], sky, tex, fast="f64", **s["kw"]).render(s["cam_pos"], s["fov"]).astype(np.float64)
print("strict f32 oracle vs f64: rmse %.3g" % np.sqrt(np.mean((ref - truth) ** 2)))
for math, comp in (("fast", False), ("fast", True), ("strict", False), ("strict", True)):
    hip = HipRenderer(s["width"], s["height"], sky, tex, math=math, **s["kw"])
    hip.render_async(s["cam_pos"], s["fov"], compaction=comp)
    L = {k: hip.read_layer(v) for k, v in (("final", 0), ("bg", 1), ("disk", 2), ("blur", 3))}
    c = hip.counters()
    print(math, "compaction", comp, "steps", c["ray_steps"], "oracle", ora.last_total_steps, "march_ms", c["march_ms"], "vgprs", c["march_vgprs"])
    for k, r in (("bg", rbg), ("disk", rdisk), ("blur", rblur), ("final", ref)):
        d = np.abs(L[k] - r).max(axis=2)
        rm = np.sqrt(np.mean((L[k].astype(np.float64) - r) ** 2, axis=(0, 1)))
        bad = np.argwhere(d > 1e-3)
        print(f"  {k}: rmse {rm} max {d.max():.4g} n>1e-3 {len(bad)} first {bad[:5].tolist()} nan {np.isnan(L[k]).sum()}")
    print("   vs f64 truth: rmse %.3g" % np.sqrt(np.mean((L["final"].astype(np.float64) - truth) ** 2)))
    hip.close()
