"""Runs tests/test_config3_gpu.py::test_run_ensemble_ragged_random_sizes_fused_equals_materialised over a range of seeds (a bug
hunt, not part of the suite).  usage: python scratch/fuzz_ensemble.py [first] [last]"""
import sys, os, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import eval as ev
import test_config3_gpu as t
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 43)
fn = getattr(t.test_run_ensemble_ragged_random_sizes_fused_equals_materialised, "__wrapped__", t.test_run_ensemble_ragged_random_sizes_fused_equals_materialised)
bad = []
for seed in range(a, b):
    try:
        fn(ev, seed)
    except Exception:          # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", traceback.format_exc().splitlines()[-1][:300], flush=True)
print("%d batches, %d failed: %s" % (b - a, len(bad), bad))
