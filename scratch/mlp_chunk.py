"""Times the tuple MLP (heads) and the point encoder at several row-chunk sizes (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd.models import BeyondCPPFShot
from bench import Cfg
dev = torch.device("cuda")
torch.manual_seed(0)
model = BeyondCPPFShot(Cfg()).to(dev).eval()
T = 64 * 20000
x = torch.randn(T, 360, device=dev)
shot = torch.randn(64 * 4096, 352, device=dev)

def run_heads(chunk):
    outs = []
    for s in range(0, T, chunk):
        outs.append(model.heads(x[s:s + chunk]))
    return outs

def run_enc(chunk):
    return [model.encode_points(shot[s:s + chunk]) for s in range(0, shot.shape[0], chunk)]

with torch.no_grad():
    for name, fn, sizes in (("heads", run_heads, [T, 640000, 320000, 160000, 80000, 40000, 20000]),
                            ("encode_points", run_enc, [262144, 65536, 32768, 16384, 8192])):
        for c in sizes:
            fn(c); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn(c)
            e1.record(); torch.cuda.synchronize()
            print(name, "chunk", c, "ms %.2f" % (e0.elapsed_time(e1) / 3), flush=True)
