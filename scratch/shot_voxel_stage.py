"""The descriptor stage alone (prepare = cells + sums + eigen, describe = SHOT352) on bench clouds, `synthetic` or `voxel2mm`
(cppf2_amd.synth.make_scene_voxel2mm: the density eval.py:185-201's voxel grid gives real inputs), both arithmetics, with hashes of
the normals and descriptors.  usage: python scratch/shot_voxel_stage.py [voxel2mm|synthetic] [reps]   (rocprofv3 --kernel-trace --stats -- python3 ...)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cppf2_amd import ops, shot, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "voxel2mm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, N = 64, 4096
dev = torch.device("cuda")
make = synth.make_scene_voxel2mm if kind == "voxel2mm" else synth.make_scene
pts = torch.from_numpy(np.concatenate([make(0, b, N)["pc"] for b in range(B)])).to(dev)
off = ops._offsets([N] * B, dev)
nrm = torch.empty((B * N, 3), device=dev)
out = torch.empty((B * N, 352), device=dev)
for arith in (("pcl", "f64") if "--both" in sys.argv else ("pcl",)):
    def prep():
        shot.prepare_device(pts, off, 0.02, 0.02, nrm, arithmetic=arith)

    def both():
        prep()
        shot.describe_device(pts, off, nrm, 0.02, out=out, nan_to_zero=True)
    both()
    torch.cuda.synchronize()
    for name, fn in (("prepare", prep), ("prepare + describe", both)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%s %s %-20s %.3f ms" % (kind, arith, name, e0.elapsed_time(e1) / reps))
    both()
    torch.cuda.synchronize()
    print(kind, arith, "normals", hashlib.sha256(nrm.cpu().numpy().tobytes()).hexdigest()[:16], "descriptors",
          hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
