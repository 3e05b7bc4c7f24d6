"""One-process-per-GPU helpers (torch.distributed over RCCL on the GPUs, gloo in CPU tests).

The path shards by independent units -- frames (video, bench) or row blocks -- so there is no
data-path collective; the only exchanges are the timing/throughput reduction of bench.py and a
barrier before rank 0 assembles a video."""
from __future__ import annotations

import os
from typing import Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def visible_device(local_rank: int) -> int:
    """The HIP device ordinal of this rank: LOCAL_RANK where the process sees every GPU of the node (torchrun's default),
    ordinal 0.. where a launcher has narrowed the view (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank: the rank's
    own GPU is then ordinal 0, and LOCAL_RANK would name a device that does not exist)."""
    import torch
    n = torch.cuda.device_count()           # counting devices does not initialise the GPU
    return local_rank if n <= 0 or local_rank < n else local_rank % n


def init(backend: str, local_rank: int = 0):
    """init_process_group with the rendezvous taken from MASTER_ADDR/MASTER_PORT (127.0.0.1 on one node)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    kw = {}
    if backend == "nccl":
        local_rank = visible_device(local_rank)
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, **kw)
    return dist


_host_group = None


def host_barrier(dist=None) -> None:
    """Barrier that keeps the GPUs idle: a gloo group on the host cores.  An RCCL barrier is a small all-reduce
    whose kernel spins on every waiting rank's GPU -- not what one wants while rank 0 times work on those GPUs
    (bench.py's row-block leg)."""
    global _host_group
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if dist.get_backend() == "gloo":
        dist.barrier()
        return
    if _host_group is None:
        _host_group = dist.new_group(backend="gloo")
    dist.barrier(group=_host_group)


def host_all_gather_bytes(payload: bytes, dist=None):
    """All-gather of one bytes object per rank over the host (gloo) group -- the exchange multigpu.TileLink needs."""
    global _host_group
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [payload]
    group = None
    if dist.get_backend() != "gloo":
        if _host_group is None:
            _host_group = dist.new_group(backend="gloo")
        group = _host_group
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, payload, group=group)
    return out


def aggregate_throughput(elapsed_s: float, units: float, dist=None, device="cpu") -> Tuple[float, float]:
    """Whole-job figures for weak scaling: (max over ranks of the elapsed time, sum over ranks of the
    units each rank processed).  Without a process group returns the inputs."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def merge_progress(temp_dir: str, world: int) -> set:
    """Union of the per-rank progress files written by drivers.render_video when world > 1."""
    import json
    done = set()
    for r in range(world):
        p = os.path.join(temp_dir, f"progress.rank{r}.json")
        if os.path.isfile(p):
            with open(p) as f:
                done |= set(json.load(f).get("completed", []))
    return done
