"""cppf_reslayer_split (split-bf16 ResLayer kernel) vs float64 and vs the library float32 path: error and time per shape.
usage: python scratch/split_bench.py [rows]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1280000
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(128, 128, False), (360, 128, True), (128, 256, True), (256, 256, False), (256, 192, True), (352, 128, True),
          (128, 64, True), (256, 128, True), (64, 64, False)]


def ref64(x, w1, b1, w0, b0, w2):
    x = x.double()
    h = torch.relu(x @ w1.double().t() + b1.double())
    skip = x if w0 is None else x @ w0.double().t() + b0.double()
    return skip + h @ w2.double().t()


def native(x, w1, b1, w0, b0, w2):
    h = torch._addmm_activation(b1, x, w1.t())
    skip = x if w0 is None else torch.addmm(b0, x, w0.t())
    return torch.addmm(skip, h, w2.t())


for k, n, proj in shapes:
    w1 = torch.randn(n, k, device=dev) / k ** 0.5
    w2 = torch.randn(n, n, device=dev) / n ** 0.5
    w0 = torch.randn(n, k, device=dev) / k ** 0.5 if proj else None
    b1 = torch.randn(n, device=dev) * 0.1
    b0 = torch.randn(n, device=dev) * 0.1 if proj else None
    wq = models.pack_split(w1, w0, w2, k)
    # accuracy on a ragged row count
    m = 3001
    x = torch.randn(m, k, device=dev)
    want = ref64(x, w1, b1, w0, b0, w2)
    got = ops.reslayer_split(x.clone(), wq, b1, b0, n)
    nat = native(x, w1, b1, w0, b0, w2)
    scale = want.abs().max().item()
    e_split = (got.double() - want).abs().max().item() / scale
    e_nat = (nat.double() - want).abs().max().item() / scale
    r_split = ((got.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    r_nat = ((nat.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    # time at full size
    x = torch.randn(rows, k, device=dev)
    out = None if not proj else torch.empty(rows, n, device=dev)
    for _ in range(2):
        ops.reslayer_split(x, wq, b1, b0, n, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ops.reslayer_split(x, wq, b1, b0, n, out=out)
    torch.cuda.synchronize()
    t_split = (time.perf_counter() - t0) / reps * 1e3
    x = torch.randn(rows, k, device=dev)
    for _ in range(2):
        native(x, w1, b1, w0, b0, w2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        native(x, w1, b1, w0, b0, w2)
    torch.cuda.synchronize()
    t_nat = (time.perf_counter() - t0) / reps * 1e3
    if (k, n, proj) == (360, 128, True):        # + the four identity layers of the tuple encoder chained in the same kernel
        rest = [(torch.randn(n, n, device=dev) / n ** 0.5, torch.randn(n, device=dev) * 0.1, torch.randn(n, n, device=dev) / n ** 0.5) for _ in range(4)]
        wqc = models.pack_split(w1, w0, w2, k, chain=[(e[0], e[2]) for e in rest])
        bias = torch.cat([b1] + [e[1] for e in rest])
        x = torch.randn(rows, k, device=dev)
        for _ in range(2):
            ops.reslayer_split(x, wqc, bias, b0, n, out=out, chain=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.reslayer_split(x, wqc, bias, b0, n, out=out, chain=4)
        torch.cuda.synchronize()
        tc = (time.perf_counter() - t0) / reps * 1e3
        fl = 2.0 * rows * (k * n * 2 + n * n + 4 * 2 * n * n)
        print("K=360 N=128 proj + 4 chained identity layers: %.3f ms (%.0f TF/s eq)" % (tc, fl / tc / 1e9), flush=True)
    flops = 2.0 * rows * (k * n * (2 if proj else 1) + n * n)
    print("K=%3d N=%3d proj=%d  max err split %.2e native %.2e | rms split %.2e native %.2e | split %.3f ms (%.0f TF/s eq) native %.3f ms (%.0f TF/s)"
          % (k, n, proj, e_split, e_nat, r_split, r_nat, t_split, flops / t_split / 1e9, t_nat, flops / t_nat / 1e9), flush=True)
