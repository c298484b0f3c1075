import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# every call through cppf2_amd._lib.load() records its symbol (tests/test_zz_stable_abi_coverage_gpu.py reads _lib.CALLED last)
os.environ.setdefault("CPPF_ABI_TRACE", "1")


BENCH2 = {}      # the bench jobs of a GPU run: {"dir", "proc", "gpus"} (tests/_bench_jobs.py runs them one after another)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_jobs():
    """The bench.py runs tests/test_multi_rank_gpu.py reads, in the order they run (each alone on the GPU):
    * counters: the counter passes of a default run -- bench.py starts rocprofv3 children of itself before it touches the GPU;
    * one_rank_16 / eight_ranks: the 8-rank launch path without 8 GPUs -- `bench.py --gpus 8 --scenes-per-gpu 2` starts eight fresh
      ranks that share GPU 0 over gloo (port allocation, LOCAL_RANK % visible GPUs, per-rank core slices and TunableOp directories);
      its 16 gathered records must equal, in global scene order, those of one rank holding all 16 scenes;
    * rccl_one_rank: `bench.py --gpus 1` with CPPF_DIST_FORCE_COLLECTIVE=1: init_process_group("nccl", device_id=...) and the path's
      all_gather (plus the bench's barrier and all_reduce) really run through RCCL, in a one-rank group;
    * one_rank_512 / eight_ranks_64: the same at the real run's geometry -- eight ranks of 64 scenes against one rank of 512;
    * refuse_two_gpus: plain `python bench.py --gpus 2` over RCCL on this one-GPU box must fail loudly;
    * two_ranks: plain `python bench.py --gpus 2` (no launcher environment): bench.py starts its two ranks itself; they share GPU 0,
      so the backend is gloo (CPPF_BENCH_BACKEND, the dry-run switch: one GPU cannot host two RCCL ranks)."""
    def job(tag, argv, scenes=4, counters=False, cpu_scenes=0, **env):
        small = ["--steps", "1", "--warmup", "0", "--scenes-per-gpu", str(scenes), "--cpu-scenes", str(cpu_scenes), "--no-reference-order",
                 "--no-native-arith", "--no-voxel-density", "--no-prior-variants", "--no-launch-power"] + ([] if counters else ["--no-counters"])
        return {"tag": tag, "argv": argv + small, "env": env}
    one = ["--no-f16x2", "--no-evidence", "--single-stream"]
    # agreement: the default run's evidence legs at a small batch -- the oracle worker pool (every scene of the batch recomputed on the
    # CPU by fresh `bench.py --oracle-worker` children started before the GPU is touched), the power / clock telemetry child with the
    # per-launch MLP loops and the library GEMM yardstick, the prior-variant loops, the voxel-density loop with ITS agreement
    agreement = {"tag": "agreement", "env": {},
                 "argv": ["--gpus", "1", "--steps", "2", "--warmup", "1", "--scenes-per-gpu", "4", "--cpu-scenes", "1", "--agreement-voxel-scenes", "2",
                          "--no-reference-order", "--no-native-arith", "--no-counters", "--no-evidence"]}
    return [agreement, job("counters", ["--gpus", "1"] + one, counters=True),
            job("one_rank_16", ["--gpus", "1"] + one, scenes=16),
            # (no --no-counters here: the ranks of a multi-rank run must decline the counter passes by themselves)
            job("eight_ranks", ["--gpus", "8"] + one, scenes=2, counters=True, CPPF_BENCH_BACKEND="gloo"),
            # the same comparison at the batch geometry of the real 8-GPU run (64 scenes per rank = 512 scenes): round 6
            job("one_rank_512", ["--gpus", "1"] + one, scenes=512),
            job("eight_ranks_64", ["--gpus", "8"] + one, scenes=64, CPPF_BENCH_BACKEND="gloo"),
            job("rccl_one_rank", ["--gpus", "1"], CPPF_DIST_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
            job("refuse_two_gpus", ["--gpus", "2"]),
            job("two_ranks", ["--gpus", "2"], CPPF_BENCH_BACKEND="gloo")]


def _wants_bench_jobs(config):
    """A GPU run that may reach tests/test_multi_rank_gpu.py (no file arguments, the tests directory, or that file)."""
    expr = config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return False
    files = [os.path.basename(a.split("::")[0]) for a in config.args]
    named = [f for f in files if f.startswith("test_") and f.endswith(".py")]
    return not named or "test_multi_rank_gpu.py" in named


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not _wants_bench_jobs(config):
        return
    try:
        import json
        import subprocess
        import tempfile
        import torch
        if torch.cuda.device_count() > 0:            # counting devices does not initialise the GPU
            d = tempfile.mkdtemp(prefix="cppf_bench2_")
            with open(os.path.join(d, "jobs.json"), "w") as f:
                json.dump(_bench_jobs(), f)
            # ONE fresh child, started before this process initialises the GPU; it runs the jobs sequentially
            BENCH2["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_bench_jobs.py"), d], cwd=ROOT)
            BENCH2["dir"] = d
            BENCH2["gpus"] = torch.cuda.device_count()
    except Exception as e:                            # pragma: no cover
        BENCH2["error"] = repr(e)


@pytest.fixture(scope="session", autouse=True)
def bench_jobs_finished():
    """The bench jobs run to completion BEFORE the first test: they never share the GPU with a test (waiting is not fork + exec)."""
    proc = BENCH2.get("proc")
    if proc is not None:
        BENCH2["runner_rc"] = proc.wait(timeout=2000)            # (eight jobs of ~15-60 s each; 240 s limit per job in the runner)
    yield


def bench_job(name):
    """(exit code, stdout, stderr) of one finished bench job, or a skip when the jobs were not started."""
    if "dir" not in BENCH2:
        pytest.skip("bench jobs were not started (%s)" % BENCH2.get("error", "not a full -m gpu run"))
    d = BENCH2["dir"]
    assert os.path.exists(os.path.join(d, "done")), "the bench job runner did not finish (rc %r)" % BENCH2.get("runner_rc")
    rc = int(open(os.path.join(d, name + ".rc")).read().split()[0])
    return rc, open(os.path.join(d, name + ".out")).read(), open(os.path.join(d, name + ".err")).read()


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
