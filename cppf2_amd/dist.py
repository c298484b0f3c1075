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


def force_collective():
    """CPPF_DIST_FORCE_COLLECTIVE=1: run the all_gather even in a one-rank group (exercises the RCCL call on a 1-GPU box)."""
    return os.environ.get("CPPF_DIST_FORCE_COLLECTIVE", "0") not in ("", "0")


INIT_TIMEOUT_S = 180.0


class DistInitError(RuntimeError):
    """The process group could not be formed: carries rank, backend and the rendezvous address in its message."""


def rendezvous_description(backend):
    return "rank %s/%s backend %s MASTER_ADDR=%s MASTER_PORT=%s LOCAL_RANK=%s" % (
        os.environ.get("RANK"), os.environ.get("WORLD_SIZE"), backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
        os.environ.get("LOCAL_RANK"))


def init(backend=None, device=None, timeout_s=None):
    """Initialises torch.distributed from the torchrun environment (no-op for a single process unless the collective is
    forced, see force_collective()).  The rendezvous and every later collective are bounded by `timeout_s` (default
    INIT_TIMEOUT_S = 180 s, CPPF_DIST_TIMEOUT_S overrides; torch's default of 10 minutes would hold a GPU box that long on a wedged
    RCCL bootstrap); a failure raises DistInitError naming the rank, the backend and MASTER_* -- the caller exits non-zero with
    it (never a re-exec)."""
    from datetime import timedelta
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    if world == 1 and not force_collective():
        return world, int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if timeout_s is None:
        timeout_s = float(os.environ.get("CPPF_DIST_TIMEOUT_S", INIT_TIMEOUT_S))
    kw = {"timeout": timedelta(seconds=float(timeout_s))}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    try:
        dist.init_process_group(backend, **kw)
    except Exception as e:      # noqa: BLE001  (store timeouts, connection refusals, RCCL bootstrap errors: all fatal here)
        raise DistInitError("init_process_group failed within %.0f s (%s): %s: %s"
                            % (timeout_s, rendezvous_description(backend), type(e).__name__, e)) from e
    return dist.get_world_size(), dist.get_rank()


_GATHER_BUFS = {}


def _gather_bufs(world, cap, device):
    """(send [cap, 160], recv [world, cap, 160]) allocated once per (world, cap, device, current stream): the step's only
    collective does not touch the allocator, and two streams gathering concurrently never share a staging buffer."""
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (world, cap, str(device), stream)
    b = _GATHER_BUFS.get(key)
    if b is None:
        if len(_GATHER_BUFS) >= 16:
            _GATHER_BUFS.clear()
        b = (torch.zeros((cap, RECORD_BYTES), dtype=torch.uint8, device=device),
             torch.empty((world, cap, RECORD_BYTES), dtype=torch.uint8, device=device))
        _GATHER_BUFS[key] = b
    return b


def gather_results(local_records, num_scenes, out=None):
    """local_records: uint8 [n_local, 160] (device or CPU).  Returns uint8 [num_scenes, 160] in global scene order
    on every rank: `out` if given (a caller that gathers every step passes a buffer it allocated once: no allocator
    traffic), else a tensor of its own -- never a view of the internal staging buffers, so the results of two gathers do not
    alias.  Shards are padded to the largest one so the all_gather is a single fixed-size collective."""
    assert local_records.dtype == torch.uint8 and local_records.shape[-1] == RECORD_BYTES
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective()):
        return local_records[:num_scenes]
    world = dist.get_world_size()
    cap = max_shard(num_scenes, world)
    dev = local_records.device
    # gloo (CPU tests, and the one-GPU dry run of the N > 1 bench) gathers host tensors; RCCL gathers in place on the device
    stage = local_records.is_cuda and dist.get_backend() == "gloo"
    src = local_records.cpu() if stage else local_records
    buf, recv = _gather_bufs(world, cap, src.device)
    buf[:src.shape[0]].copy_(src)
    if src.is_cuda:
        dist.all_gather_into_tensor(recv, buf)
    else:
        dist.all_gather(list(recv.unbind(0)), buf)
    if num_scenes == world * cap:
        res = recv.view(num_scenes, RECORD_BYTES)
        if out is not None:
            out.copy_(res)
            return out
        return res.to(dev) if stage else res.clone()
    if out is None:
        out = torch.empty((num_scenes, RECORD_BYTES), dtype=torch.uint8, device=src.device)
    for r in range(world):
        lo, hi = shard(num_scenes, r, world)
        out[lo:hi].copy_(recv[r, :hi - lo])
    return out.to(dev) if stage else out
