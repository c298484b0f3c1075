"""Is cppf_reslayer_split power-bound?  Same launch on random and on all-zero operands (the switching activity of the
matrix pipe depends on the data), with wall-clock per launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
dev = torch.device("cuda:0")
rows = 1280000
for k, n, proj in [(256, 256, False), (360, 128, True)]:
    for zero in (False, True, False):
        torch.manual_seed(0)
        mk = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, device=dev))
        w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
        w0 = mk(n, k) / k ** 0.5 if proj else None
        b1 = mk(n) * 0.1
        b0 = mk(n) * 0.1 if proj else None
        wq = models.pack_split(w1, w0, w2, k)
        x = mk(rows, k)
        out = torch.empty(rows, n, device=dev) if proj else None
        for _ in range(3):
            ops.reslayer_split(x, wq, b1, b0, n, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.reslayer_split(x, wq, b1, b0, n, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        flops = 2.0 * rows * (k * n * (2 if proj else 1) + n * n)
        print("K=%d N=%d proj=%d %s: %.3f ms  %.0f TF/s eq (%.2f PF/s of bf16 MFMA)" % (k, n, proj, "zeros " if zero else "random", dt, flops / dt / 1e9, 6 * flops / dt / 1e12), flush=True)
