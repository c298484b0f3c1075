"""N > 1 path on CPU: world_size-2/3 gloo processes shard scenes and exchange the 160-byte records with the one
all_gather of the path.  No GPU compute: records are synthesised from the global scene id."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cppf2_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys  # noqa: E402


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


def test_init_times_out_with_rank_backend_and_address_in_the_error():
    """Round 6 hardening: a rendezvous that cannot complete (rank 1 of 2, nobody listening on the master port) fails within the
    given timeout -- not torch's 10 minutes -- with a DistInitError naming rank, backend and MASTER_*."""
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cppf2_amd import dist\n"
            "try:\n"
            "    dist.init(backend='gloo', timeout_s=4)\n"
            "except dist.DistInitError as e:\n"
            "    print('DISTINIT', e); sys.exit(4)\n"
            "sys.exit(0)\n" % ROOT)
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 4, p.stdout[-2000:]
    assert time.time() - t0 < 60
    line = [l for l in p.stdout.splitlines() if l.startswith("DISTINIT")][0]
    for word in ("rank 1/2", "backend gloo", "MASTER_ADDR=127.0.0.1", "MASTER_PORT=%d" % port, "within 4 s"):
        assert word in line, (word, line)


def test_run_ranks_reports_every_rank_and_stops_the_survivors():
    """benchlib.launch.run_ranks: one rank fails -> the others (which would wait in a collective) are stopped by PID, every rank's
    exit status and stderr tail are printed, the launcher returns the failing status -- within seconds, not a collective timeout."""
    import io
    import time
    from cppf2_amd.benchlib import launch
    good = [sys.executable, "-c", "import sys, time; print('rank alive', file=sys.stderr, flush=True); time.sleep(120)"]
    bad = [sys.executable, "-c", "import sys, time; time.sleep(1); print('RCCL bootstrap: connection refused', file=sys.stderr); sys.exit(3)"]
    buf = io.StringIO()
    t0 = time.time()
    rc = launch.run_ranks([good, bad, good], 1, 29999, timeout_s=60, out=buf)
    assert rc == 3 and time.time() - t0 < 30
    text = buf.getvalue()
    assert "multi-rank run FAILED: rank 1 exited with status 3" in text
    assert "---- rank 0: exit stopped by the launcher" in text and "rank alive" in text
    assert "---- rank 1: exit 3" in text and "RCCL bootstrap: connection refused" in text and "---- rank 2:" in text
    # a run that exceeds its limit is stopped and reported too
    buf = io.StringIO()
    t0 = time.time()
    rc = launch.run_ranks([good, good], 1, 29999, timeout_s=2, out=buf)
    assert rc != 0 and time.time() - t0 < 30 and "the run exceeded 2 s" in buf.getvalue()
    # a good run: status 0, rank 0's stderr forwarded
    ok = [sys.executable, "-c", "import sys; print('note from rank', file=sys.stderr)"]
    buf = io.StringIO()
    assert launch.run_ranks([ok, ok], 1, 29999, timeout_s=60, out=buf) == 0 and "note from rank" in buf.getvalue()
