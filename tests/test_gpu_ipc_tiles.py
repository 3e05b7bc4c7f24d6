"""Row blocks with one PROCESS per tile (bhr_tile_export / _connect / _render: HIP IPC memory handles + shared-memory
counters): three processes share the test box's card, each renders its block of the frame, rank 0 ends up with the
frame -- equal to the frame of one context."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, world, gather, frames=3, math="strict", size=None):
    shm = "bhr_test_" + uuid.uuid4().hex[:12]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if size:
        env["BHR_TEST_SIZE"] = f"{size[0]}x{size[1]}"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_tile_worker.py"), str(tmp_path), str(k), str(world), shm,
                               gather, str(frames), math], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for k in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}:\n{outs[k][-3000:]}"
    return np.load(os.path.join(tmp_path, "frame.npy")), sum(int(np.load(os.path.join(tmp_path, f"steps{k}.npy"))[0]) for k in range(world))


@pytest.mark.parametrize("world,gather", [(3, "peer"), (2, "peer_u8")])
def test_one_process_per_tile_equals_one_context(tmp_path, world, gather, hip_lib):
    from bhr_amd import HipRenderer
    got, steps = _run(tmp_path, world, gather)
    s = scenes.SCENES["default"]
    full = HipRenderer(640, 360, scenes.analytic_skybox(), scenes.noisy_disk(), frame_slots=1, **s["kw"])
    ref = full.render(s["cam_pos"], s["fov"])
    assert steps == full.counters()["ray_steps"]
    # row blocks are bit-identical to the whole frame: the exact-f32 V pass adds a column's rows in ascending order
    # whatever the tiling, the halo rows come through the IPC mappings as they were written
    if gather == "peer":
        np.testing.assert_array_equal(got, ref)
    else:
        np.testing.assert_array_equal(got, full.read_final_u8())
    full.close()


def test_one_process_per_tile_hybrid_with_the_split_post_pass(tmp_path, hip_lib):
    """960x544 (bloom radius 19), hybrid: every process marches its block's strict and fast tile lists and runs the split-f16 V
    pass over halo rows that arrived through IPC handles -- the same bits as one hybrid context."""
    from bhr_amd import HipRenderer
    got, steps = _run(tmp_path, 3, "peer", math="hybrid", size=(960, 544))
    s = scenes.SCENES["default"]
    full = HipRenderer(960, 544, scenes.analytic_skybox(), scenes.noisy_disk(), frame_slots=1, math="hybrid", **s["kw"])
    ref = full.render(s["cam_pos"], s["fov"])
    assert steps == full.counters()["ray_steps"]
    np.testing.assert_array_equal(got, ref)
    full.close()
