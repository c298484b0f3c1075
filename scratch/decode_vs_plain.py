"""The logit head's output layer (256 -> 192 projection ResLayer) with and without the bin-draw epilogue, and the neighbouring
shapes for comparison: ms per launch at the bench's 1.28 M rows.  usage: python scratch/decode_vs_plain.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
rows = 64 * 20000


def layer(k, n, proj, chain=0):
    w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
    w0 = mk(n, k) / k ** 0.5 if proj else None
    rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
    return models.pack_split(w1, w0, w2, k, chain=rest), mk((1 + chain) * n) * 0.1, (mk(n) * 0.1 if proj else None)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = mk(rows, 256)
uni = torch.rand(rows, 6, generator=g).to(dev)
prior = mk(rows, 192)
bins = torch.empty((rows, 6), dtype=torch.int32, device=dev)
out192 = torch.empty((rows, 192), device=dev)
out256 = torch.empty((rows, 256), device=dev)
wq, b1, b0 = layer(256, 192, True)
flop = lambda k, n, proj, chain=0: (2.0 * n * k * (2 if proj else 1) + 2.0 * n * n + chain * 4.0 * n * n) * rows / 1e9
for name, fn, gf in (("256->192 + draw (prior)", lambda: ops.reslayer_split_decode(x, wq, b1, b0, uni, prior=prior, bins=bins), flop(256, 192, True)),
                     ("256->192 + draw (no prior)", lambda: ops.reslayer_split_decode(x, wq, b1, b0, uni, bins=bins), flop(256, 192, True)),
                     ("256->192 plain (writes [T,192])", lambda: ops.reslayer_split(x, wq, b1, b0, 192, out=out192), flop(256, 192, True))):
    ms = timeit(fn)
    print("%-34s %.3f ms  %.1f ns/GFLOP-row  %.0f TFLOP/s" % (name, ms, 1e6 * ms / gf, gf / ms))
wq2, b12, _ = layer(256, 256, False, 1)
ms = timeit(lambda: ops.reslayer_split(x, wq2, b12, None, 256, out=out256, chain=1))
print("%-34s %.3f ms  %.0f TFLOP/s" % ("256 identity x2 (chain 1)", ms, flop(256, 256, False, 1) / ms))
wq3, b13, b03 = layer(256, 256, True, 0)
ms = timeit(lambda: ops.reslayer_split(x, wq3, b13, b03, 256, out=out256))
print("%-34s %.3f ms  %.0f TFLOP/s" % ("256->256 projection", ms, flop(256, 256, True) / ms))
