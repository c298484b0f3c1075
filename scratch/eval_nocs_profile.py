"""Host-side profile of eval.main(data="nocs") on a rendered set in the REAL275 layout (the test suite's renderer): wall time per
instance, cProfile of the loop.  usage: python scratch/eval_nocs_profile.py [images]"""
import sys, os, time, tempfile, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import torch
import eval as ev
from test_entry_points_gpu import _write_nocs_fixture_rendered
images = int(sys.argv[1]) if len(sys.argv) > 1 else 12
root = tempfile.mkdtemp()
recs = _write_nocs_fixture_rendered(root, images=images)
n = sum(len(r["pred_class_ids"]) for r in recs)
kw = dict(data="nocs", log_dir=os.path.join(root, "log"), data_root=os.path.join(root, "real_test"), num_pairs=50000, num_rots=180,
          opt=True, batch_instances=16, seed=0)
ev.main(**kw)                                   # warm-up: kernels, weight packing
torch.cuda.synchronize(); t0 = time.perf_counter()
ev.main(**kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("eval.main on %d images / %d detections: %.2f s = %.1f ms per detection" % (images, n, dt, 1e3 * dt / n))
pr = cProfile.Profile(); pr.enable(); ev.main(**kw); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(30)
st.print_callers("method 'to' of")
st.sort_stats("tottime").print_stats(25)
