import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shot_oracle as S
from cppf2_amd import shot, ops
n = 6000
rng = np.random.RandomState(n); v = rng.randn(n, 3)
pc = (v / np.linalg.norm(v, axis=1, keepdims=True) * (rng.rand(n, 1) ** (1 / 3)) * 0.03 + 0.5).astype(np.float32)
pts = torch.as_tensor(pc).cuda()
hs, hn, hrf = shot.compute_device(pts, ops._offsets([n], pts.device), 0.02, 0.02, want_rf=True)
hs, hn, hrf = hs.cpu().numpy(), hn.cpu().numpy(), hrf.cpu().numpy()
os_, on, orf, d = S.compute_ex(pc, 0.02, 0.02)
err = np.abs(hs - os_).max(1)
bad = np.where(err >= 2e-5)[0]
print("bad", bad, "rf err", np.abs(hrf - orf).max(1)[bad])
for i in bad:
    a, b = (hs[i] * d[i, 6]).reshape(32, 11), (os_[i] * d[i, 6]).reshape(32, 11)      # un-normalised weights
    diff = a - b
    idx = np.argwhere(np.abs(diff) > 0.02)
    print("row", i, "wrap %.3g edge %.3g nbrs %d" % (d[i, 5], d[i, 8], d[i, 7]), "sum diff %.4f" % diff.sum(), "sum|diff| %.4f" % np.abs(diff).sum())
    for r, c in idx[:12]:
        print("    sector %2d slot %2d hip %.4f oracle %.4f diff %+.4f" % (r, c, a[r, c], b[r, c], diff[r, c]))
