"""cppf_reslayer_split16 (f16x2) vs cppf_reslayer_split (bf16x3) vs the library float32 path: error against float64 and time.
usage: python scratch/split16_bench.py [rows]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1280000
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(256, 256, False, 1), (128, 256, True, 2), (256, 192, True, 0), (360, 128, True, 4), (128, 128, False, 0), (128, 64, True, 0), (352, 128, True, 4)]

def ref64(x, layers):
    x = x.double()
    for (w1, b1, w0, b0, w2) in layers:
        h = torch.relu(x @ w1.double().t() + b1.double())
        skip = x if w0 is None else x @ w0.double().t() + b0.double()
        x = skip + h @ w2.double().t()
    return x

def native(x, layers):
    for (w1, b1, w0, b0, w2) in layers:
        h = torch._addmm_activation(b1, x, w1.t())
        skip = x if w0 is None else torch.addmm(b0, x, w0.t())
        x = torch.addmm(skip, h, w2.t())
    return x

def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for k, n, proj, chain in shapes:
    layers = []
    kk = k
    for li in range(1 + chain):
        w1 = torch.randn(n, kk, device=dev) / kk ** 0.5
        w2 = torch.randn(n, n, device=dev) / n ** 0.5
        w0 = torch.randn(n, kk, device=dev) / kk ** 0.5 if (proj and li == 0) else None
        b1 = torch.randn(n, device=dev) * 0.1
        b0 = torch.randn(n, device=dev) * 0.1 if (proj and li == 0) else None
        layers.append((w1, b1, w0, b0, w2)); kk = n
    w1, b1, w0, b0, w2 = layers[0]
    rest = [(l[0], l[4]) for l in layers[1:]]
    b1s = torch.cat([l[1] for l in layers])
    wq3 = models.pack_split(w1, w0, w2, k, chain=rest)
    sc = models.f16_scale(w1, w0, w2, *[w for p in rest for w in p])
    wq2 = models.pack_split(w1, w0, w2, k, chain=rest, arith="f16x2", scale=sc)
    x = torch.randn(3001, k, device=dev)
    want = ref64(x, layers); scale = want.abs().max().item()
    g3 = ops.reslayer_split(x.clone(), wq3, b1s, b0, n, chain=chain)
    g2 = ops.reslayer_split16(x.clone(), wq2, b1s * sc, None if b0 is None else b0 * sc, n, sc, chain=chain)
    nat = native(x, layers)
    err = lambda t: ((t.double() - want).abs().max().item() / scale, ((t.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item())
    e3, e2, en = err(g3), err(g2), err(nat)
    xb = torch.randn(rows, k, device=dev)
    out = None if not proj else torch.empty(rows, n, device=dev)
    t3 = timeit(lambda: ops.reslayer_split(xb, wq3, b1s, b0, n, out=out, chain=chain))
    xb = torch.randn(rows, k, device=dev)
    t2 = timeit(lambda: ops.reslayer_split16(xb, wq2, b1s * sc, None if b0 is None else b0 * sc, n, sc, out=out, chain=chain))
    fl = 2.0 * rows * (k * n * (2 if proj else 1) + n * n + chain * 2 * n * n)
    print("K=%3d N=%3d proj=%d chain=%d scale 2^%d | max err bf16x3 %.2e f16x2 %.2e library %.2e | rms %.2e %.2e %.2e | ms bf16x3 %.3f f16x2 %.3f (%.0f / %.0f TF/s f32-eq)"
          % (k, n, proj, chain, int(torch.log2(torch.tensor(sc)).item()), e3[0], e2[0], en[0], e3[1], e2[1], en[1], t3, t2, fl / t3 / 1e9, fl / t2 / 1e9), flush=True)
