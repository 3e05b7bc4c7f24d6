#!/usr/bin/env python3
"""One 1/8 tile of the 8k frame timed end to end on ONE GPU (march -> H -> halo pull -> V chunks -> rows landed in the
gather buffer on tile 0) against its march alone, for both schedules of bhr_group_render; plus the whole frame in 8
blocks on this device.  The other seven tiles keep the buffers of a previous full render (bhr_group_render_subset).

usage: python tools/exp_tile_tail.py [--tiles 8] [--workload 8k] [--math strict] [--out gpurun_out/tile_tail.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bhr_amd import multigpu, workloads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--workload", default="8k")
    ap.add_argument("--math", default=None)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_tail.json"))
    a = ap.parse_args()
    res = bench.tile_tail_leg(bench.WORKLOADS[a.workload], a.tiles, math=a.math, reps=a.reps, verbose=True, schedules=("pipelined", "serial"),
                              gathers=("peer_u8", "peer"))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_tile"}, indent=1))


if __name__ == "__main__":
    main()
