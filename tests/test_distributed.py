"""N > 1 path on the CPU: two gloo ranks exercise the sharding + reduction logic that bench.py and
the video driver use under torchrun (one process per GPU; backend nccl == RCCL there)."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bhr_amd  # noqa: F401
    from bhr_amd import distributed as D
    from bhr_amd.multigpu import frames_of_rank, row_blocks
    assert D.env_rank() == (rank, world, rank)
    d = D.init("gloo")
    # weak scaling: every rank renders its own frames; whole-job value = sum(units) / max(time)
    elapsed, units = D.aggregate_throughput(1.0 + rank, 150e6 * (rank + 1), d)
    assert elapsed == float(world) and units == 150e6 * world * (world + 1) / 2
    # frame shards are disjoint and complete
    mine = list(frames_of_rank(37, rank, world))
    gathered = [None] * world
    d.all_gather_object(gathered, mine)
    assert sorted(f for g in gathered for f in g) == list(range(37))
    # row blocks: rank k owns block k
    blocks = row_blocks(1080, world)
    assert blocks[rank][1] - blocks[rank][0] in (1080 // world, 1080 // world + 1)
    # per-rank progress files merge into the full set
    with open(os.path.join(tmp, f"progress.rank{rank}.json"), "w") as f:
        json.dump({"params": {}, "completed": mine}, f)
    d.barrier()
    D.host_barrier(d)                       # bench.py's GPU-idle barrier (gloo group beside the data backend)
    if rank == 0:
        assert D.merge_progress(tmp, world) == set(range(37))
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_sharding_and_reduction(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def test_single_process_passthrough():
    import bhr_amd  # noqa: F401
    from bhr_amd import distributed as D
    assert D.aggregate_throughput(2.5, 7.0, None) == (2.5, 7.0)
