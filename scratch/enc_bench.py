"""encode_tuples_shot at the bench workload (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops
B, N, T = int(os.environ.get("B", 64)), 4096, 20000
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
pts = torch.randn((B * N, 3), generator=g).to(dev)
nrm = torch.nn.functional.normalize(torch.randn((B * N, 3), generator=g), dim=-1).to(dev)
feat = torch.randn((B * N, 64), generator=g).to(dev)
idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
pt_off, tup_off = ops._offsets([N] * B, dev), ops._offsets([T] * B, dev)
x = ops.encode_tuples_shot(pts, idx, feat, nrm, pt_off, tup_off)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    x = ops.encode_tuples_shot(pts, idx, feat, nrm, pt_off, tup_off)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("CPPF_ENC_BX", os.environ.get("CPPF_ENC_BX"), "encode ms %.3f  out GB/s %.0f  checksum %.6f" % (ms, x.numel() * 4 / 1e9 / (ms / 1e3), float(x[::1000].sum())))
