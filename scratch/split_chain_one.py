"""The tuple encoder's first five layers as one cppf_reslayer_split launch (360 -> 128 projection + 4 chained identity
layers) in a loop, for rocprofv3: python scratch/split_chain_one.py [rows] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1280000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
k, n = 360, 128
w1, w0, w2 = torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, n, device=dev) / n ** 0.5
rest = [(torch.randn(n, n, device=dev) / n ** 0.5, torch.randn(n, n, device=dev) / n ** 0.5) for _ in range(4)]
wq = models.pack_split(w1, w0, w2, k, chain=rest)
bias = torch.randn(5 * n, device=dev) * 0.1
b0 = torch.randn(n, device=dev) * 0.1
x = torch.randn(rows, k, device=dev)
out = torch.empty(rows, n, device=dev)
for _ in range(reps):
    ops.reslayer_split(x, wq, bias, b0, n, out=out, chain=4)
torch.cuda.synchronize()
