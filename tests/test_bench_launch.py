"""bench.py's launch contract on a box without GPUs: `--gpus N` never silently runs fewer ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "CPPF_BENCH_BACKEND", "CPPF_DIST_FORCE_COLLECTIVE")


def _run(argv, **env_add):
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    env.update(env_add)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=300)


def test_plain_multi_gpu_invocation_fails_loudly_without_the_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and "{" not in r.stdout


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "1"], WORLD_SIZE="2", RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="1")
    assert r.returncode == 2 and "{" not in r.stdout
