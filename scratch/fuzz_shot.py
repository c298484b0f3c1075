"""Runs tests/test_gpu_edges.py::test_shot_random_clouds_vs_oracle over a range of seeds (a bug hunt, not part of the suite).
usage: python scratch/fuzz_shot.py [first] [last]"""
import sys, os, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import test_gpu_edges as t
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 66)
bad = []
for seed in range(a, b):
    for arith in ("f64", "pcl"):
        try:
            t.test_shot_random_clouds_vs_oracle(seed, arith)
        except Exception:          # noqa: BLE001
            bad.append((seed, arith))
            print("seed", seed, arith, "FAILED:", traceback.format_exc().splitlines()[-1][:300], flush=True)
print("%d clouds x 2 arithmetics, %d failed: %s" % (b - a, len(bad), bad))
