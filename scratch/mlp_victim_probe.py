"""The other direction: is the MLP kernel's output affected by small-kernel workgroups of another stream sharing its CUs?"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cppf2_amd import models, ops
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.Step(args, 0, 1, dev)
st.run(); torch.cuda.synchronize()
pipe = st.pipe
idx = ops.sample_tuples(4096, 20000, 5, 0, tuple(range(64)), dev)
side = torch.cuda.Stream()
g = torch.Generator(device="cpu").manual_seed(1)
w1 = (torch.randn(256, 256, generator=g) / 16).to(dev); w2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
wq = models.pack_split(w1, None, w2, 256); b1 = (torch.randn(256, generator=g) * 0.1).to(dev)
x = torch.randn(400000, 256, device=dev)
ref = ops.reslayer_split(x, wq, b1, None, 256, out=torch.empty_like(x)); torch.cuda.synchronize()
out = torch.empty_like(x)
from cppf2_amd import shot as shotmod
for name, fn in (("none", lambda: None), ("rot_bins", lambda: [pipe.rot_bins(st.pts, idx) for _ in range(12)]),
                 ("vote_center", lambda: [pipe.vote_center(st.pts, idx) for _ in range(4)]),
                 ("shot", lambda: [shotmod.prepare_device(st.pts, pipe.pt_off, 0.02, 0.02, st.normal) for _ in range(4)])):
    bad = 0
    for rep in range(10):
        with torch.cuda.stream(side):
            fn()
        ops.reslayer_split(x, wq, b1, None, 256, out=out)
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print("MLP 256-wide beside %-12s: %d / 10 outputs differ from the solo run" % (name, bad), flush=True)
