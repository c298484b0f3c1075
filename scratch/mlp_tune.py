"""TunableOp experiment for the MLP GEMMs (not part of the product)."""
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
def run():
    model.encode_points(shot); model.heads(x)
with torch.no_grad():
    import time
    t0 = time.time(); run(); torch.cuda.synchronize(); print("first pass (tuning) s %.1f" % (time.time() - t0), flush=True)
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        print("ms %.2f" % (e0.elapsed_time(e1) / 3), flush=True)
