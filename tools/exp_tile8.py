"""Row-block leg rehearsal: the 8k frame in n row blocks, all on device 0 (BHR_TILE_DEVICES=0,0,...): what the tiled
path costs over one context when the devices are not real (launch / exchange overhead, load balance of the cut).
Usage: python tools/exp_tile8.py [n ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    os.environ["BHR_TILE_DEVICES"] = ",".join(["0"] * n)
    t = bench.tile_leg(bench.WORKLOADS["8k"], n, frames=8)
    print(n, "tiles:", round(t["ms_per_frame"], 2), "ms/frame; march per tile", t["tile_ms"]["march"], "blocks", t["row_blocks"], flush=True)
