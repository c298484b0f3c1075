import numpy as np, torch, sys
sys.path.insert(0,'.')
from cppf2_amd import ops
g=dict(np.load('tests/golden/small.npz'))
tr,rot=ops.generate_target_pairs(g['small_scaled'],[0,1,0],[0,0,1],[1,0,0])
d=tr!=g['small_tr0']
print('mismatch', d.sum(0), 'of', tr.shape)
i=np.argwhere(d)
for r,c in i[:8]:
    print(r,c, tr[r,c], g['small_tr0'][r,c], tr[r,c]-g['small_tr0'][r,c])
