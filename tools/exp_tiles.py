"""Row-block load balance of the 8k config (configs[3]): march time of each of N tiles, measured one
after another on one device -> predicted N-GPU efficiency = mean / max.  Usage: exp_tiles.py [workload] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bhr_amd import HipRenderer
from bhr_amd.workloads import make_scene
from bhr_amd.multigpu import row_blocks

name = sys.argv[1] if len(sys.argv) > 1 else "8k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.WORKLOADS[name]
full, sky, tex, _ = make_scene(wl, n_stars=2000)
cam, fov = wl["cam_pos"], wl["fov"]
for _ in range(2):
    full.render_async(cam, fov)
c = full.counters()
print(f"{name}: whole frame march {c['march_ms']:.3f} ms bloom {c['bloom_ms']:.3f} ms", flush=True)
full.close()
kw = dict(step_size=wl["step_size"], r_max=10.0, disk_tilt=wl["disk_tilt"], anti_alias=wl["anti_alias"])


def measure(blocks):
    t = []
    for rows in blocks:
        r = HipRenderer(wl["width"], wl["height"], sky, tex, rows=rows, **kw)
        for _ in range(3):
            r.render_async(cam, fov)
        c = r.counters()
        t.append((c["march_ms"], c["bloom_ms"], c["ray_steps"]))
        r.close()
    return np.array(t)


blocks = row_blocks(wl["height"], n)
t = measure(blocks)
print("uniform blocks:", [f"{a:.2f}" for a in t[:, 0]], f"efficiency {t[:, 0].mean() / t[:, 0].max():.3f}", flush=True)
print("ray-steps share:", [f"{a / t[:, 2].sum():.3f}" for a in t[:, 2]])
from bhr_amd.multigpu import balanced_row_blocks, probe_row_costs
per_row, band = probe_row_costs(wl["width"], wl["height"], cam, fov, **kw)
for fixed in (0.0, 0.1, 0.2):
    blocks = balanced_row_blocks(wl["height"], n, per_row, band, fixed_cost_per_row=fixed * float(per_row.mean()))
    t = measure(blocks)
    print(f"balanced (fixed {fixed}):", [b[1] - b[0] for b in blocks], [f"{a:.2f}" for a in t[:, 0]],
          f"efficiency {t[:, 0].mean() / t[:, 0].max():.3f}, max {t[:, 0].max():.2f} ms, "
          f"vs one GPU {c['march_ms'] / (n * t[:, 0].max()):.3f}", flush=True)
