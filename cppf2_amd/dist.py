"""Scene sharding over ranks + the single collective of the path.

Scenes are independent units: rank r of W owns a contiguous block of global scene ids, runs the whole path on
them, and the fixed 160-byte records are exchanged with ONE all_gather (backend "nccl" = RCCL over xGMI on the GPU
box; "gloo" in the CPU tests).  Global scene id keys the RNG streams, so results do not depend on W.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_BYTES = 160


def shard(num_scenes, rank, world):
    """Contiguous block [lo, hi) of global scene ids owned by `rank`; blocks differ by at most one scene."""
    base, rem = divmod(int(num_scenes), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def max_shard(num_scenes, world):
    return (int(num_scenes) + int(world) - 1) // int(world)


def init(backend=None, device=None):
    """Initialises torch.distributed from the torchrun environment (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world, int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist.get_world_size(), dist.get_rank()


def gather_results(local_records, num_scenes):
    """local_records: uint8 [n_local, 160] (device or CPU).  Returns uint8 [num_scenes, 160] in global scene order
    on every rank.  Shards are padded to the largest one so the all_gather is a single fixed-size collective."""
    assert local_records.dtype == torch.uint8 and local_records.shape[-1] == RECORD_BYTES
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_records[:num_scenes]
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = max_shard(num_scenes, world)
    dev = local_records.device
    # gloo (CPU tests, and the one-GPU dry run of the N > 1 bench) gathers host tensors; RCCL gathers in place on the device
    stage = local_records.is_cuda and dist.get_backend() == "gloo"
    src = local_records.cpu() if stage else local_records
    buf = torch.zeros((cap, RECORD_BYTES), dtype=torch.uint8, device=src.device)
    buf[:src.shape[0]] = src
    out = torch.empty((world, cap, RECORD_BYTES), dtype=torch.uint8, device=src.device)
    if src.is_cuda:
        dist.all_gather_into_tensor(out, buf)
    else:
        dist.all_gather(list(out.unbind(0)), buf)
    parts = []
    for r in range(world):
        lo, hi = shard(num_scenes, r, world)
        parts.append(out[r, :hi - lo])
    res = torch.cat(parts, 0)
    return res.to(dev) if stage else res
