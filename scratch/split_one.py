"""One shape of cppf_reslayer_split in a loop (for rocprofv3): python scratch/split_one.py K N proj [rows] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
k, n, proj = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 1280000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
w1 = torch.randn(n, k, device=dev) / k ** 0.5
w2 = torch.randn(n, n, device=dev) / n ** 0.5
w0 = torch.randn(n, k, device=dev) / k ** 0.5 if proj else None
b1 = torch.randn(n, device=dev) * 0.1
b0 = torch.randn(n, device=dev) * 0.1 if proj else None
wq = models.pack_split(w1, w0, w2, k)
x = torch.randn(rows, k, device=dev)
out = torch.empty(rows, n, device=dev) if proj else None
for _ in range(reps):
    ops.reslayer_split(x, wq, b1, b0, n, out=out)
torch.cuda.synchronize()
