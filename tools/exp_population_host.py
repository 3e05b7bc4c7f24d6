"""Host time of the per-frame texture calls with an idle device (every call preceded by a device sync): what of the video
loop's per-frame time is host work inside the library / the binding.  Usage: python tools/exp_population_host.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import drivers
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math="hybrid")
factories = drivers.init_lifecycle_system(r, r.dtex_h, r.dtex_w, seed=42)
acc = {"tick": [], "generate_background": [], "accumulate_entity_layer": [], "compose": [], "render_async": [], "sync_after": []}
for f in range(300):
    t = f * 0.1
    r.sync()
    t0 = time.perf_counter()
    for fa in factories.values():
        fa.tick(now=t, dt=0.1)
    t1 = time.perf_counter(); r.generate_background(t=t)
    t2 = time.perf_counter(); r.accumulate_entity_layer(factories, now=t)
    t3 = time.perf_counter(); r.compose_interactive_texture(solo_idx=-1)
    t4 = time.perf_counter(); r.render_async([6, 0, 0.5], 90)
    t5 = time.perf_counter(); r.sync()
    t6 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        acc[k].append(v * 1e3)
for k, v in acc.items():
    v = np.array(v[50:])
    print(f"{k:26s} median {np.median(v):.3f} ms  mean {v.mean():.3f}  max {v.max():.3f}")
ents = {k: len(f.alive_entities) for k, f in factories.items()}
print("alive entities", ents, "texture", r.dtex_h, r.dtex_w)
r.close()
