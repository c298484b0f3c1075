import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


BENCH2 = {}      # handles of the bench jobs started in pytest_configure (GPU runs only)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CPPF_BENCH_BACKEND",
               "CPPF_DIST_FORCE_COLLECTIVE")


# a job with delay > 0 is started by this stub (which never touches the GPU): it sleeps, runs bench.py as its child and exits with
# the child's code -- the counter passes of the "counters" job then measure before the multi-rank jobs load the GPU
_DELAYED = ("import subprocess, sys, time\n"
            "time.sleep(float(sys.argv[1]))\n"
            "sys.exit(subprocess.call([sys.executable] + sys.argv[2:]))\n")


def _start_bench(tmpdir, tag, argv, scenes=4, counters=False, delay=0.0, **env_add):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    env.update(env_add)
    out = open(os.path.join(tmpdir, tag + ".out"), "w")
    err = open(os.path.join(tmpdir, tag + ".err"), "w")
    small = ["--steps", "1", "--warmup", "0", "--scenes-per-gpu", str(scenes), "--cpu-scenes", "0", "--no-reference-order",
             "--no-native-arith"] + ([] if counters else ["--no-counters"])
    cmd = [os.path.join(ROOT, "bench.py")] + argv + small
    cmd = [sys.executable] + cmd if delay <= 0 else [sys.executable, "-c", _DELAYED, str(delay)] + cmd
    return (subprocess.Popen(cmd, env=env, stdout=out, stderr=err, cwd=ROOT), out.name, err.name)


def _start_bench_jobs(tmpdir):
    """FRESH child processes, started here, before this process initialises the GPU -- a process that has done so must not
    fork+exec on this pool -- and collected by tests/test_multi_rank_gpu.py:
    * two_ranks: plain `python bench.py --gpus 2` (no launcher environment): bench.py starts its two ranks itself; they share
      GPU 0, so the backend is gloo (CPPF_BENCH_BACKEND, the dry-run switch: one GPU cannot host two RCCL ranks);
    * rccl_one_rank: `bench.py --gpus 1` with CPPF_DIST_FORCE_COLLECTIVE=1: init_process_group("nccl", device_id=...) and
      the path's all_gather (plus the bench's barrier and all_reduce) really run through RCCL, in a one-rank group;
    * refuse_two_gpus: plain `python bench.py --gpus 2` over RCCL on this one-GPU box must fail loudly;
    * eight_ranks / one_rank_16: the 8-rank launch path without 8 GPUs -- `bench.py --gpus 8 --scenes-per-gpu 2` starts eight
      fresh ranks that share GPU 0 over gloo (port allocation, LOCAL_RANK % visible GPUs, per-rank core slices and TunableOp
      directories); its 16 gathered records must equal, in global scene order, those of one rank holding all 16 scenes."""
    return {
        # the counter passes of a default run: bench.py starts rocprofv3 children of itself before it touches the GPU
        "counters": _start_bench(tmpdir, "counters", ["--gpus", "1", "--no-f16x2", "--no-evidence", "--single-stream"], counters=True),
        "eight_ranks": _start_bench(tmpdir, "eight_ranks", ["--gpus", "8", "--no-f16x2", "--no-evidence", "--single-stream"], scenes=2,
                                    delay=25.0, CPPF_BENCH_BACKEND="gloo"),
        "one_rank_16": _start_bench(tmpdir, "one_rank_16", ["--gpus", "1", "--no-f16x2", "--no-evidence", "--single-stream"], scenes=16),
        "two_ranks": _start_bench(tmpdir, "two_ranks", ["--gpus", "2"], delay=15.0, CPPF_BENCH_BACKEND="gloo"),
        "rccl_one_rank": _start_bench(tmpdir, "rccl_one_rank", ["--gpus", "1"], CPPF_DIST_FORCE_COLLECTIVE="1",
                                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
        "refuse_two_gpus": _start_bench(tmpdir, "refuse_two_gpus", ["--gpus", "2"]),
    }


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import tempfile
            import torch
            if torch.cuda.device_count() > 0:            # counting devices does not initialise the GPU
                BENCH2["jobs"] = _start_bench_jobs(tempfile.mkdtemp(prefix="cppf_bench2_"))
                BENCH2["gpus"] = torch.cuda.device_count()
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
