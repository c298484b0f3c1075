"""rot_bins beside different hogs / rot_bins variants: which combination changes the counts?"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppf2_amd import _lib
if os.environ.get('CPPF_PROBE_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['CPPF_PROBE_LIB'])
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
def layer(k, n):
    w1 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev); w2 = (torch.randn(n, n, generator=g) / n ** 0.5).to(dev)
    return w1, w2, models.pack_split(w1, None, w2, k), torch.zeros(n, device=dev)
w1a, w2a, wq256, b256 = layer(256, 256)
w1b, w2b, wq128, b128 = layer(128, 128)
w1c, w2c, wq64, b64 = layer(64, 64)
x256 = torch.randn(400000, 256, device=dev); x128 = torch.randn(800000, 128, device=dev); x64 = torch.randn(800000, 64, device=dev)
big = torch.randn(8192, 8192, device=dev)
def proj(k, n):
    w1 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev); w0 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev); w2 = (torch.randn(n, n, generator=g) / n ** 0.5).to(dev)
    return models.pack_split(w1, w0, w2, k), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
wqp256, bp1, bp0 = proj(128, 256)
wqp192, bq1, bq0 = proj(256, 192)
o256 = torch.empty(800000, 256, device=dev); o192 = torch.empty(400000, 192, device=dev)
wq192i = layer(192, 192); x192 = torch.randn(400000, 192, device=dev)
only = os.environ.get("CPPF_PROBE_HOGS")
hogs = {"none": lambda: None,
        "proj128->256": lambda: ops.reslayer_split(x128, wqp256, bp1, bp0, 256, out=o256),
        "proj256->192": lambda: ops.reslayer_split(x256, wqp192, bq1, bq0, 192, out=o192),
        "id192": lambda: ops.reslayer_split(x192, wq192i[2], wq192i[3], None, 192),
        "split256_small": lambda: ops.reslayer_split(x256[:20000], wq256, b256, None, 256),
        "split256": lambda: ops.reslayer_split(x256, wq256, b256, None, 256),
        "split128": lambda: ops.reslayer_split(x128, wq128, b128, None, 128),
        "split64": lambda: ops.reslayer_split(x64, wq64, b64, None, 64),
        "reslayer128_f32": lambda: ops.reslayer128_(x128, w1b, b128, w2b),
        "torch_gemm": lambda: torch.mm(big, big),
        "vote_center": None}
victims = {"lut": lambda: pipe.rot_bins(st.pts, idx)}
for vname, vfn in victims.items():
    vfn(); torch.cuda.synchronize()
    ref = pipe.counts.clone()
    for hname, hfn in hogs.items():
        if hfn is None or (only and hname not in only.split(',')): continue
        bad = 0
        for rep in range(10):
            with torch.cuda.stream(side):
                hfn()
            vfn()
            torch.cuda.synchronize()
            bad += int(not torch.equal(pipe.counts, ref))
        print("victim %-6s hog %-16s: %d / 10 differ" % (vname, hname, bad), flush=True)
