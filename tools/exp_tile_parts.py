"""Every row block of the 8k frame rendered alone, its march bracketed inside the frame (BHR_GROUP_TIME_MARCH):
frame = lead-in + march + post-pass.  usage: python tools/exp_tile_parts.py [math]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bhr_amd import multigpu, workloads

math = sys.argv[1] if len(sys.argv) > 1 else "hybrid"
wl = bench.WORKLOADS["8k"]
tiles, blocks, note = workloads.make_tiles(wl, [0] * 8, math=math)
cam, fov = wl["cam_pos"], wl["fov"]
t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="pipelined")
for k in range(8):
    live = [1 if q == k else 0 for q in range(8)]
    rows = []
    for tm in (False, True):
        for _ in range(3):
            multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="pipelined", live=live, time_march=tm)
        fr, ma = [], []
        for _ in range(8):
            multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="pipelined", live=live, time_march=tm)
            c = tiles[k].counters()
            fr.append(c["frame_ms"]); ma.append(c["march_ms"])
        rows.append((np.median(fr), np.median(ma)))
    for _ in range(3):
        tiles[k].render_async(cam, fov, skip_bloom=True)
    al = []
    for _ in range(8):
        tiles[k].render_async(cam, fov, skip_bloom=True)
        al.append(tiles[k].counters()["march_ms"])
    info = tiles[k].hybrid_info() if math == "hybrid" else {}
    print(f"tile {k} rows {blocks[k]}: frame {rows[0][0]:.3f}  | bracketed: frame {rows[1][0]:.3f} march {rows[1][1]:.3f} rest {rows[1][0] - rows[1][1]:.3f} | march alone (no packed stores) {np.median(al):.3f}  strict tiles {info.get('strict_tiles')}", flush=True)
multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="pipelined")
for x in tiles:
    x.close()
