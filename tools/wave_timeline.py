"""Occupancy over time of one march launch from per-wave stamps (BHR_WAVE_STAMPS=<file>, csrc/march.hip):
how long the launch runs at full occupancy, how long its ragged end is, what the last waves were doing.
Needs a diagnostic build of the library: make -C black-hole-renderer_amd/csrc EXTRA=-DBHR_WAVE_STAMPS_BUILD=1 (after touching march.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd  # noqa
from bhr_amd import workloads

path = "/tmp/wave_stamps.bin"
wl = dict(width=1920, height=1080, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=0.0, anti_alias="disabled")
math = sys.argv[1] if len(sys.argv) > 1 else None                 # strict (default) / fast: one launch; hybrid dumps its LAST launch (the fast list)
r, _, _, _ = workloads.make_scene(wl, frame_slots=1, math=math)
for _ in range(30):
    r.render_async(wl["cam_pos"], wl["fov"])
r.sync()
os.environ["BHR_WAVE_STAMPS"] = path
r.render_async(wl["cam_pos"], wl["fov"]); r.sync()
del os.environ["BHR_WAVE_STAMPS"]
c = r.counters(); r.close()
w = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
steps = (w[:, 2] & np.uint64((1 << 40) - 1)).astype(np.int64); flushes = (w[:, 2] >> np.uint64(40)).astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base                         # 100 MHz ticks = 10 ns
T = t1.max()
print(f"march_ms (events) {c['march_ms']:.3f}; stamps span {T / 100:.1f} us; {len(w)} waves; wave-steps {steps.sum() / 64:.0f}")
dur = (t1 - t0) / 100.0
print(f"wave lifetime us: mean {dur.mean():.1f} median {np.median(dur):.1f} p99 {np.quantile(dur, .99):.1f} max {dur.max():.1f}; "
      f"per wave-step ns: median {np.median(dur * 1e3 / np.maximum(steps / 64, 1)):.0f}")
# waves alive over time
ev = np.zeros(T + 2, dtype=np.int64); np.add.at(ev, t0, 1); np.add.at(ev, t1 + 1, -1); alive = np.cumsum(ev)[:T + 1]
cap = alive.max()
for frac in (0.99, 0.9, 0.5, 0.25, 0.1):
    below = np.nonzero(alive < frac * cap)[0]; below = below[below > T // 2]
    print(f"  occupancy falls below {frac:4.0%} of its peak ({cap} waves) at {below[0] / 100 if len(below) else T / 100:7.1f} us")
print(f"  mean occupancy over the launch: {alive.mean() / cap:.3f} of peak; time-integrated waves x us: {alive.sum() / 100:.0f}")
last = np.argsort(t1)[-8:]
for k in last:
    print(f"  late wave slot {k}: start {t0[k] / 100:7.1f} end {t1[k] / 100:7.1f} us, max-lane steps ~{steps[k]} summed, flushes {flushes[k]}")
# start ramp
starts = np.sort(t0)
print(f"  first 5120 waves started by {starts[min(5119, len(starts) - 1)] / 100:.1f} us; 50 % of all waves started by {starts[len(starts) // 2] / 100:.1f} us")
print("  t(us)  alive  started_in_bin  ended_in_bin")
B = 2500   # 25 us bins
for b in range(0, T + 1, B):
    s_in = int(((t0 >= b) & (t0 < b + B)).sum()); e_in = int(((t1 >= b) & (t1 < b + B)).sum())
    print(f"  {b / 100:6.0f} {int(alive[min(b + B // 2, T)]):6d} {s_in:8d} {e_in:8d}")
# per SE/CU residency at the 450 us mark: which CUs are under-filled
hw = w[:, 3].astype(np.int64)
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
mark = int(T // 2)
mid = (t0 <= mark) & (t1 >= mark)
key = se[mid] * 32 + sh[mid] * 16 + cu[mid]
cnt = np.bincount(key, minlength=256)
print(f"  waves alive at {mark / 100:.0f} us per (SE,SH,CU) id: min", cnt[cnt > 0].min() if (cnt > 0).any() else 0, "median", int(np.median(cnt[cnt > 0])), "max", cnt.max(), "ids seen", (cnt > 0).sum())
