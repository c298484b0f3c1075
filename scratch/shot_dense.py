"""SHOT stage on clouds at the density of a 2 mm voxel grid (what eval.py:185-201 produces): ~300 neighbours inside the 2 cm support,
against the bench's synthetic clouds (~90).  usage: python scratch/shot_dense.py [radius_m height_m]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, shot
B, N = 64, 4096
rad = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
hgt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
dev = torch.device("cuda")
rng = np.random.default_rng(0)
clouds = []
for b in range(B):
    th = rng.uniform(0, 2 * np.pi, N); z = rng.uniform(0, hgt, N)
    p = np.stack([rad * np.cos(th), z, rad * np.sin(th) + 0.8], 1)
    p = np.round(p / 0.002) * 0.002 + rng.uniform(-2e-4, 2e-4, (N, 3))        # voxel-grid-like spacing
    clouds.append(p.astype(np.float32))
pts = torch.from_numpy(np.concatenate(clouds)).to(dev)
off = ops._offsets([N] * B, dev)
nrm = torch.empty((B * N, 3), device=dev)
out = torch.empty((B * N, 352), device=dev)
d = torch.cdist(pts[:N], pts[:N])
print("neighbours inside 2 cm: mean %.0f, max %d" % ((d < 0.02).sum(1).float().mean().item(), (d < 0.02).sum(1).max().item()))
for arith in ("pcl", "f64"):
    def prep():
        shot.prepare_device(pts, off, 0.02, 0.02, nrm, arithmetic=arith)
    def both():
        prep(); shot.describe_device(pts, off, nrm, 0.02, out=out, nan_to_zero=True)
    both(); torch.cuda.synchronize()
    for name, fn in (("prepare", prep), ("prepare + describe", both)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print("%s %-20s %.3f ms" % (arith, name, e0.elapsed_time(e1) / 10))
    both(); torch.cuda.synchronize()
    print(arith, "normals", hashlib.sha256(nrm.cpu().numpy().tobytes()).hexdigest()[:12], "descriptors", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12])
