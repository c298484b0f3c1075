"""Can the VALU-bound HIP kernels hide under the MLP's GEMMs when issued on a second stream?  (experiment)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth, shot
from cppf2_amd.models import BeyondCPPFShot
from bench import Cfg
dev = torch.device("cuda")
B, N, T = 32, 4096, 20000
pts = torch.from_numpy(np.concatenate([synth.make_scene(0, b, N)["pc"] for b in range(B)])).to(dev)
off = ops._offsets([N] * B, dev)
model = BeyondCPPFShot(Cfg()).to(dev).eval()
x = torch.randn(B * T, 360, device=dev)
nrm = torch.empty((B * N, 3), device=dev); desc = torch.empty((B * N, 352), device=dev)
def hip():
    for _ in range(4):
        shot.prepare_device(pts, off, 0.02, 0.02, nrm); shot.describe_device(pts, off, nrm, 0.02, out=desc)
def mlp():
    with torch.no_grad():
        model.heads(x, lazy_scale=True)
s2 = torch.cuda.Stream()
def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2): hip()
    mlp()
    torch.cuda.current_stream().wait_stream(s2)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a, b, c = t(mlp), t(hip), t(both)
print("mlp %.2f ms  hip %.2f ms  sum %.2f  concurrent %.2f  hidden %.0f%% of hip" % (a, b, a + b, c, 100 * (a + b - c) / b))
