"""DINO-model tuple encode (a3') at bench size: gather-add of per-point products vs gather + Linear over the concatenation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd.models import BeyondCPPFDino
from bench import Cfg
dev = torch.device("cuda")
B, N, T = 64, 4096, 20000
m = BeyondCPPFDino(Cfg()).to(dev).eval()
pts = torch.randn(B * N, 3, device=dev)
desc = torch.nn.functional.normalize(torch.randn(B * N, 1024, device=dev), dim=-1)
idx = (torch.randint(0, N, (B * T, 5), device=dev) + (torch.arange(B, device=dev).repeat_interleave(T) * N)[:, None]).int()
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def new():
    with torch.no_grad(): return m.prepare_tuple_inputs(pts, desc, idx)
def old():
    with torch.enable_grad(): return m.prepare_tuple_inputs(pts, desc, idx)
a = new(); b = old().detach()
print("max diff", float((a - b).abs().max()))
print("gather-add %.2f ms   gather + Linear(1280->256) %.2f ms" % (t(new), t(old)))
