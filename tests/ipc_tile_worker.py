"""Worker of tests/test_gpu_ipc_tiles.py: one rank of the one-process-per-tile row-block path.
usage: python tests/ipc_tile_worker.py <dir> <rank> <world> <shm name> <gather> <frames> [math]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    d, rank, world, shm, gather, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    math = sys.argv[7] if len(sys.argv) > 7 else "strict"
    from bhr_amd import HipRenderer, multigpu, scenes
    s = scenes.SCENES["default"]
    W, H = 640, 360                                        # R = 12: tiles of 360 / world rows are thicker than the halo
    if os.environ.get("BHR_TEST_SIZE"):                     # e.g. 960x544: radius 19, the split-f16 post-pass under fast / hybrid
        W, H = (int(v) for v in os.environ["BHR_TEST_SIZE"].split("x"))
    cuts = [round(H * k / world / 8) * 8 for k in range(world)] + [H]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    r = HipRenderer(W, H, sky, tex, rows=(cuts[rank], cuts[rank + 1]), frame_slots=1, math=math, **s["kw"])
    link = multigpu.TileLink(r, rank, world, multigpu.file_exchange(d, rank, world), shm, gather=gather)
    for f in range(frames):
        link.render(s["cam_pos"], s["fov"])
    steps = r.counters()["ray_steps"]
    if rank == 0:
        np.save(os.path.join(d, "frame.npy"), link.read_gathered())
    np.save(os.path.join(d, f"steps{rank}.npy"), np.array([steps], dtype=np.int64))
    link.close()
    r.close()


if __name__ == "__main__":
    main()
