"""Runs tests/test_gpu_edges.py::test_vote_center_slab_cut_random_batches over a range of seeds (a bug hunt, not part of the suite).
usage: python scratch/fuzz_vote_slabs.py [first] [last]"""
import sys, os, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import test_gpu_edges as t
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 110)
bad = []
for seed in range(a, b):
    try:
        t.test_vote_center_slab_cut_random_batches(seed)
    except Exception:          # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", traceback.format_exc().splitlines()[-1][:300], flush=True)
print("%d batches, %d failed: %s" % (b - a, len(bad), bad))
