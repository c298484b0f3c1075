"""128-wide ResLayer: the fused matrix-core kernel against the two library GEMMs it replaces (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import ops
rows = int(os.environ.get("ROWS", 64 * 20000))
g = torch.Generator(device="cpu").manual_seed(0)
w1 = (torch.randn((128, 128), generator=g) * 0.1).cuda(); w2 = (torch.randn((128, 128), generator=g) * 0.1).cuda()
b1 = torch.randn((128,), generator=g).cuda()
x = torch.randn((rows, 128), generator=g).cuda()

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def lib():
    h = torch._addmm_activation(b1, x, w1.t())
    x.addmm_(h, w2.t())

with torch.no_grad():
    t_lib = timeit(lib)
    t_k = timeit(lambda: ops.reslayer128_(x, w1, b1, w2))
fl = 2 * 2 * rows * 128 * 128
print("rows %d: library 2 GEMMs %.3f ms (%.0f TF/s) | fused kernel %.3f ms (%.0f TF/s, %.0f GB/s)" % (rows, t_lib, fl / t_lib / 1e9, t_k, fl / t_k / 1e9, rows * 128 * 4 * 2 / t_k / 1e6))
