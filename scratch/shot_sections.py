"""Times normals + SHOT352 at the bench size with each library in argv (default: the product library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppf2_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from cppf2_amd import ops, synth, shot
B, N = 64, 4096
dev = torch.device("cuda")
pts = torch.from_numpy(np.concatenate([synth.make_scene(0, b, N)["pc"] for b in range(B)])).to(dev)
off = ops._offsets([N] * B, dev)
nrm = torch.empty((B * N, 3), device=dev); out = torch.empty((B * N, 352), device=dev)
def prep(): shot.prepare_device(pts, off, 0.02, 0.02, nrm)
def desc(): shot.describe_device(pts, off, nrm, 0.02, out=out, nan_to_zero=True)
prep(); desc(); torch.cuda.synchronize()
res = {}
for name, fn in (("prepare", lambda: prep()), ("describe", lambda: desc())):
    if name == "describe": prep()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn() if name == "prepare" else (desc())
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / 20
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product", "prepare %.3f ms  describe %.3f ms" % (res["prepare"], res["describe"]),
      "checksum %.6f" % float(out.double().sum()), "sha", __import__("hashlib").sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
