"""Runs the bench.py jobs of the GPU suite ONE AFTER ANOTHER, each as a fresh child process, and records their output.

Started by tests/conftest.py in pytest_configure -- before the pytest process initialises the GPU (a process that has done so must
not fork + exec on this pool) -- and waited for by a session fixture before the first test runs: no job shares the GPU with
another job or with a test, so nothing in tests/test_multi_rank_gpu.py depends on load or timing.  This runner never imports torch
and never touches the GPU itself.

    python tests/_bench_jobs.py <dir>      # reads <dir>/jobs.json = [{"tag", "argv", "env"}...]; writes <tag>.out/.err/.rc, then done
"""
import json
import os
import subprocess
import sys
import time

LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CPPF_BENCH_BACKEND",
              "CPPF_DIST_FORCE_COLLECTIVE")


def main(d):
    jobs = json.load(open(os.path.join(d, "jobs.json")))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for job in jobs:
        env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
        env.update(job.get("env", {}))
        t0 = time.time()
        with open(os.path.join(d, job["tag"] + ".out"), "w") as out, open(os.path.join(d, job["tag"] + ".err"), "w") as err:
            try:
                rc = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + job["argv"], env=env, stdout=out, stderr=err,
                                    cwd=root, timeout=job.get("timeout", 240)).returncode
            except subprocess.TimeoutExpired:
                rc = -9
        with open(os.path.join(d, job["tag"] + ".rc"), "w") as f:
            f.write("%d %.1f" % (rc, time.time() - t0))
    open(os.path.join(d, "done"), "w").write("ok")


if __name__ == "__main__":
    main(sys.argv[1])
