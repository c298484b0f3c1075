"""DINO-model forward at bench size, stage by stage: library-only table form (round 4) against the row form of rounds 2-3, and the
SHOT model's first launch from slot tables against its gathered form.  usage: python scratch/dino_step.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
from cppf2_amd.models import BeyondCPPFDino, BeyondCPPFShot, fused_stack
from bench import Cfg
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, T = 4096, 20000
torch.manual_seed(0)
m = BeyondCPPFDino(Cfg()).to(dev).eval()
ms = BeyondCPPFShot(Cfg()).to(dev).eval()
pts = torch.randn(B * N, 3, device=dev) * 0.05
nrm = torch.nn.functional.normalize(torch.randn(B * N, 3, device=dev), dim=-1)
desc = torch.nn.functional.normalize(torch.randn(B * N, 1024, device=dev), dim=-1)
idx = torch.randint(0, N, (B * T, 5), device=dev).int()
pt_off, tup_off = ops._uniform_offsets(N, B, dev), ops._uniform_offsets(T, B, dev)
gl = idx + (torch.arange(B, device=dev).repeat_interleave(T) * N)[:, None].int()
u = torch.rand(B * T, 6, device=dev)
bins = torch.empty((B * T, 6), dtype=torch.int32, device=dev)


def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n


with torch.no_grad():
    fold = m.first_layer_fold(5)
    pp = m.transform_points(desc)
    tab = fold.tables(pp)
    heads, gidx = ops.encode_tuples_coord_heads(pts, idx, pt_off, tup_off)
    print("desc_transform (linear_split 1024->256)  %.3f ms" % t(lambda: m.transform_points(desc)))
    print("slot tables    (linear_split 256->1280)  %.3f ms" % t(lambda: fold.tables(pp)))
    print("coord heads                              %.3f ms" % t(lambda: ops.encode_tuples_coord_heads(pts, idx, pt_off, tup_off)))
    print("DINO heads_from_tuples (draw, lazy)      %.3f ms" % t(lambda: m.heads_from_tuples(pts, desc, idx, pt_off, tup_off, lazy_scale=True, decode=(u, None, bins))))
    print("  same, tables given                     %.3f ms" % t(lambda: m.heads_from_tuples(pts, desc, idx, pt_off, tup_off, lazy_scale=True, decode=(u, None, bins), tables=tab)))
    print("DINO row form: prepare_tuple_inputs      %.3f ms" % t(lambda: m.prepare_tuple_inputs(pts, desc, gl)))
    x = m.prepare_tuple_inputs(pts, desc, gl)
    print("DINO row form: heads (draw, eager scale) %.3f ms" % t(lambda: m.heads(x, decode=(u, None, bins))))
    del x
    # SHOT model
    feat = torch.randn(B * N, 64, device=dev)
    sf = ms.first_layer_fold(64, 5)
    print("SHOT slot tables (linear_split 64->1280) %.3f ms" % t(lambda: sf.tables(feat)))
    print("SHOT heads_from_tuples gathered          %.3f ms" % t(lambda: ms.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off, lazy_scale=True, decode=(u, None, bins))))
    print("SHOT heads_from_tuples slot tables       %.3f ms" % t(lambda: ms.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off, lazy_scale=True, decode=(u, None, bins), sum_tables=True)))
