import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth, shot
B, N = 64, 4096
dev = torch.device("cuda")
pts = torch.from_numpy(np.concatenate([synth.make_scene(0, b, N)["pc"] for b in range(B)])).to(dev)
off = ops._offsets([N] * B, dev)
nrm = shot.normals_device(pts, off, 0.02)
out = shot.descriptors_device(pts, off, nrm, 0.02)
torch.cuda.synchronize()
for name, fn in (("normals", lambda: shot.normals_device(pts, off, 0.02, out=nrm)), ("desc", lambda: shot.descriptors_device(pts, off, nrm, 0.02, out=out))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "ms %.3f" % (e0.elapsed_time(e1) / 10))
print("checksum", float(torch.nan_to_num(out).double().sum()))
