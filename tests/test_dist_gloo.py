"""N > 1 path on CPU: world_size-2/3 gloo processes shard scenes and exchange the 160-byte records with the one
all_gather of the path.  No GPU compute: records are synthesised from the global scene id."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cppf2_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _record(scene_id):
    rng = np.random.RandomState(scene_id)
    return rng.randint(0, 256, D.RECORD_BYTES).astype(np.uint8)


def _worker(rank, world, port, num_scenes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    w, r = D.init(backend="gloo")
    assert (w, r) == (world, rank)
    lo, hi = D.shard(num_scenes, rank, world)
    local = torch.from_numpy(np.stack([_record(s) for s in range(lo, hi)]) if hi > lo
                             else np.zeros((0, D.RECORD_BYTES), np.uint8))
    allr = D.gather_results(local, num_scenes).numpy().copy()
    # the send / receive buffers are allocated once and reused: a second gather of different records is still right
    again = D.gather_results(255 - local, num_scenes).numpy().copy()
    assert np.array_equal(again, 255 - allr) and len(D._GATHER_BUFS) == 1
    q.put((rank, allr))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,num_scenes", [(2, 8), (2, 7), (3, 5)])
def test_gather_results_gloo(world, num_scenes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_scenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_record(s) for s in range(num_scenes)])
    for rank, arr in got:
        assert arr.shape == want.shape and np.array_equal(arr, want), rank


def _worker_forced(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CPPF_DIST_FORCE_COLLECTIVE="1")
    os.environ.pop("RANK", None)
    os.environ.pop("WORLD_SIZE", None)
    assert D.init(backend="gloo") == (1, 0) and torch.distributed.is_initialized()
    local = torch.from_numpy(np.stack([_record(s) for s in range(5)]))
    allr = D.gather_results(local, 5)
    q.put((allr.data_ptr() != local.data_ptr(), allr.numpy().copy()))     # went through the collective's receive buffer
    torch.distributed.destroy_process_group()


def test_forced_collective_in_a_one_rank_group():
    """CPPF_DIST_FORCE_COLLECTIVE=1 (how the GPU suite runs the RCCL branch on a one-GPU box): group of one, real all_gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(_free_port(), q))
    p.start()
    through_collective, arr = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and through_collective
    assert np.array_equal(arr, np.stack([_record(s) for s in range(5)]))


def test_shard_covers_every_scene_once():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 64, 511, 512):
            seen = []
            for r in range(world):
                lo, hi = D.shard(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo <= D.max_shard(n, world)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
