"""The tuple encoders' first launch alone at bench size: table-fed (sumgather) for the DINO and the SHOT model, x-tile gathered for
the SHOT model.  usage: python scratch/first_launch.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
from cppf2_amd.models import BeyondCPPFDino, BeyondCPPFShot
from bench import Cfg
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, T = 4096, 20000
torch.manual_seed(0)
m, ms = BeyondCPPFDino(Cfg()).to(dev).eval(), BeyondCPPFShot(Cfg()).to(dev).eval()
pts = torch.randn(B * N, 3, device=dev) * 0.05
nrm = torch.nn.functional.normalize(torch.randn(B * N, 3, device=dev), dim=-1)
idx = torch.randint(0, N, (B * T, 5), device=dev).int()
pt_off, tup_off = ops._uniform_offsets(N, B, dev), ops._uniform_offsets(T, B, dev)


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n


def stream(net, fold, k_in):
    plan, _ = models._fused_plan(net.tuple_encoder)
    e = plan[0]
    chain = [(q[0].t(), q[4].t()) for q in plan[1:5]]
    b1 = e[1] if fold is None or fold.b1_add is None else e[1] + fold.b1_add
    b0 = e[3] if fold is None or fold.b0_add is None else e[3] + fold.b0_add
    w1, w0 = (e[0].t(), e[2].t()) if fold is None else (fold.w1_heads, fold.w0_heads)
    wq = models.pack_split(w1, w0, e[4].t(), k_in, chain=chain)
    return wq, torch.cat([b1] + [q[1] for q in plan[1:5]]).contiguous(), b0.contiguous()


with torch.no_grad():
    fold = m.first_layer_fold(5)
    tab = torch.randn(B * N, 1280, device=dev)
    heads, gidx = ops.encode_tuples_coord_heads(pts, idx, pt_off, tup_off)
    wq, b1, b0 = stream(m, fold, 32)
    print("DINO first launch, slot tables   %.3f ms" % t(lambda: ops.reslayer_split_sumgather(heads, gidx, tab, wq, b1, b0, 128, chain=4)))
    sf = ms.first_layer_fold(64, 5)
    h2, g2 = ops.encode_tuples_shot_heads(pts, idx, nrm, pt_off, tup_off)
    wq2, b12, b02 = stream(ms, sf, 40)
    print("SHOT first launch, slot tables   %.3f ms" % t(lambda: ops.reslayer_split_sumgather(h2, g2, tab, wq2, b12, b02, 128, chain=4)))
    feat = torch.randn(B * N, 64, device=dev)
    wq3, b13, b03 = stream(ms, None, 360)
    print("SHOT first launch, x-tile gather %.3f ms" % t(lambda: ops.reslayer_split_gather(h2, g2, feat, wq3, b13, b03, 128, chain=4)))
