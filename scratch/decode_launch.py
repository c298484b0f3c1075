"""The logit head's output layer with the bin draw (cppf_reslayer_split_decode) alone at bench size.  usage: python scratch/decode_launch.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
from cppf2_amd.models import BeyondCPPFShot
from bench import Cfg
dev = torch.device("cuda")
T = 64 * 20000
torch.manual_seed(0)
ms = BeyondCPPFShot(Cfg()).to(dev).eval()
x = torch.randn(T, 256, device=dev)
u = torch.rand(T, 6, device=dev)
prior = torch.randn(T, 192, device=dev)
bins = torch.empty((T, 6), dtype=torch.int32, device=dev)
with torch.no_grad():
    plan, _ = models._fused_plan(ms.logit_encoder)
    e = plan[2]
    wq = models.pack_split(e[0].t(), e[2].t(), e[4].t(), 256)
    b1, b0 = e[1].contiguous(), e[3].contiguous()
    f = lambda: ops.reslayer_split_decode(x, wq, b1, b0, u, prior=prior, bins=bins)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("256 -> 192 + bin draw: %.3f ms   bins sha %s" % (e0.elapsed_time(e1) / 10, __import__("hashlib").sha256(bins.cpu().numpy().tobytes()).hexdigest()[:12]))
    # the same layer with the logits written instead (no draw), and the draw without a prior
    out = torch.empty((T, 192), device=dev)
    for name, g in (("256 -> 192, logits written (no draw)", lambda: ops.reslayer_split(x, wq, b1, b0, 192, out=out)),
                    ("256 -> 192 + bin draw, no prior", lambda: ops.reslayer_split_decode(x, wq, b1, b0, u, bins=bins))):
        g(); g(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10): g()
        e1.record(); torch.cuda.synchronize()
        print("%s: %.3f ms" % (name, e0.elapsed_time(e1) / 10))
