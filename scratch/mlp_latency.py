"""Tuple MLP latency at small batch sizes (B scenes x 20 000 tuples): split vs native arithmetic."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models
from cppf2_amd.config import load_config
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.BeyondCPPFShot(load_config("config", "config", ["category=bottle"])).to(dev).eval()
for B in (1, 2, 4, 8, 16, 64):
    x = torch.randn(B * 20000, 360, device=dev)
    res = {}
    for mode in ("split", "native"):
        models.MLP_ARITH = mode
        with torch.no_grad():
            for _ in range(3):
                net.heads(x.clone(), lazy_scale=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                net.heads(x.clone(), lazy_scale=True)
            torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 10 * 1e3
    print("B=%2d  rows %7d  split %.3f ms  native %.3f ms" % (B, B * 20000, res["split"], res["native"]), flush=True)
