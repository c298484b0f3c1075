import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops
g = dict(np.load('tests/golden/small.npz'))
for R, T in ((36, 512), (36, 64), (180, 512)):
    cs, sn = ops.rotation_table(R)
    trig = (g['small_cos'], g['small_sin']) if R == 36 else None
    a, _ = ops.vote_center(g['small_pc'], g['small_tr0'][:T], 2e-3, g['small_idx'][:T, :2], R, trig=trig, mode=1)
    b, _ = ops.vote_center(g['small_pc'], g['small_tr0'][:T], 2e-3, g['small_idx'][:T, :2], R, trig=trig, mode=2)
    d = a - b
    print(R, T, 'sum1', a.sum(), 'sum2', b.sum(), 'cells diff', (d != 0).sum(), 'missing', d[d < 0].sum(), 'extra', d[d > 0].sum(), a.shape)
