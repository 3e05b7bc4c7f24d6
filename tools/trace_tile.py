"""One row block of the 8k frame rendered alone (its neighbours resting), for a kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/trace_tile.py 7 [math]
then `python tools/trace_tile.py --timeline OUT` prints the kernels of the last frame: start (us from the first), duration, queue."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeline(d, frames=1):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # frames are separated by host gaps; the last `frames` groups of kernels at least 200 us apart
    groups, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 200_000:
            groups.append(cur)
            cur = []
        cur.append(b)
    groups.append(cur)
    for g in groups[-frames:]:
        t0 = int(g[0]["Start_Timestamp"])
        print(f"--- frame of {len(g)} kernels, {(max(int(r['End_Timestamp']) for r in g) - t0) / 1e3:.1f} us")
        for r in g:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            print(f"  {s / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', '?'):>9} wg {r.get('Workgroup_Size', '?'):>4}  {r['Kernel_Name'][:70]}")


def main():
    if sys.argv[1] == "--timeline":
        return timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    import bench
    from bhr_amd import multigpu, workloads
    k = int(sys.argv[1])
    math = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
    wl = bench.WORKLOADS["8k"]
    tiles, blocks, note = workloads.make_tiles(wl, [0] * 8, math=math)
    cam, fov = wl["cam_pos"], wl["fov"]
    import time
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="pipelined")
    live = [1 if q == k else 0 for q in range(8)]
    for _ in range(6):
        multigpu.group_render(tiles, cam, fov, gather="peer_u8", schedule="serial", live=live)
        time.sleep(0.002)
    print(blocks, tiles[k].counters()["frame_ms"])
    for x in tiles:
        x.close()


if __name__ == "__main__":
    main()
