"""Why a fuzz cloud of scratch/fuzz_shot.py fails: rows / normals that differ and the oracle's own margins there.
usage: python scratch/fuzz_shot_diag.py seed arithmetic"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import numpy as np
import test_gpu_edges as t
from cppf2_amd import shot
seed, arith = int(sys.argv[1]), sys.argv[2]
pc, rn, rs = t._random_cloud(np.random.RandomState(700 + seed))
hs, hn = shot.compute(pc, rn, rs, arithmetic=arith)
os_, on, d = t._shot_oracle(pc, rn, rs, arith)
hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
print("n", pc.shape[0], "rn", rn, "rs", rs, "centre", pc.mean(0), "extent", pc.max(0) - pc.min(0))
dn = np.abs(hn - on).max(1)
bad = np.nonzero(dn > t.NORMAL_TOL[arith])[0]
print("normals differing:", bad.size, "of", (~np.isnan(on[:, 0])).sum())
for i in bad[:8]:
    v = pc[i] / np.linalg.norm(pc[i])
    nb = ((pc - pc[i]) ** 2).sum(1) < rn * rn
    print("  point", i, "nbrs", int(nb.sum()), "hip", hn[i], "oracle", on[i], "dot", float(hn[i] @ on[i]), "n.view hip %.2e oracle %.2e" % (float(hn[i] @ v), float(on[i] @ v)))
ok = ~np.isnan(os_).any(1)
err = np.abs(hs - os_).max(1)
badr = np.nonzero(ok & (err >= 2e-5))[0]
print("descriptor rows differing >= 2e-5:", badr.size, "of", int(ok.sum()))
for i in badr[:8]:
    print("  row", i, "err %.3g" % err[i], "margins d5 %.3g d8 %.3g" % (d[i, 5], d[i, 8]), "all d", np.array2string(d[i], precision=3))
