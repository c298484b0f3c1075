import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


BENCH2 = {}      # handles of the two-rank bench job started in pytest_configure (GPU runs only)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _start_two_rank_bench(tmpdir):
    """bench.py --gpus 2 as two FRESH child processes sharing GPU 0 (backend gloo: one GPU cannot host two RCCL ranks).
    Started here, before this process initialises the GPU -- a process that has done so must not fork+exec on this
    pool -- and collected by tests/test_multi_rank_gpu.py."""
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CPPF_BENCH_BACKEND="gloo")
        out = open(os.path.join(tmpdir, "bench2_rank%d.out" % rank), "w")
        err = open(os.path.join(tmpdir, "bench2_rank%d.err" % rank), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                                        "--warmup", "0", "--scenes-per-gpu", "4", "--cpu-scenes", "0"],
                                       env=env, stdout=out, stderr=err, cwd=ROOT), out.name, err.name))
    return procs


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import tempfile
            import torch
            if torch.cuda.device_count() > 0:            # counting devices does not initialise the GPU
                BENCH2["procs"] = _start_two_rank_bench(tempfile.mkdtemp(prefix="cppf_bench2_"))
        except Exception as e:                            # pragma: no cover
            BENCH2["error"] = repr(e)


@pytest.fixture(scope="session")
def small():
    return dict(np.load(os.path.join(GOLDEN, "small.npz")))


@pytest.fixture(scope="session")
def full_summary():
    import json
    with open(os.path.join(GOLDEN, "full_summary.json")) as f:
        return json.load(f)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
