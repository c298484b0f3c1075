"""The bench's SHOT stage alone (prepare = cells, covariance sums, eigen-solves; describe = SHOT352) with hashes of its outputs,
for before / after comparisons of kernel changes.  usage: python scratch/shot_stage.py [pcl|f64]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth, shot
B, N = 64, 4096
arith = sys.argv[1] if len(sys.argv) > 1 else None
dev = torch.device("cuda")
pts = torch.from_numpy(np.concatenate([synth.make_scene(0, b, N)["pc"] for b in range(B)])).to(dev)
off = ops._offsets([N] * B, dev)
nrm = torch.empty((B * N, 3), device=dev)
out = torch.empty((B * N, 352), device=dev)


def prep():
    shot.prepare_device(pts, off, 0.02, 0.02, nrm, arithmetic=arith)


def both():
    prep()
    shot.describe_device(pts, off, nrm, 0.02, out=out, nan_to_zero=True)


both(); torch.cuda.synchronize()
for name, fn in (("prepare", prep), ("prepare + describe", both)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-20s %.3f ms" % (name, e0.elapsed_time(e1) / 20))
both(); torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
print("normals", h(nrm), "descriptors", h(out))
