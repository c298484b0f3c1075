"""Centre vote through GLOBAL atomics (vote_center_global_kernel, cppf_vote_center mode 2 -- what grids above the LDS-slab budget
take) against the LDS-slab path (mode 1) on the same inputs, at a given object extent.
usage: python scratch/vote_global.py <mode 1|2> <extent_x> <extent_y> <extent_z> [B] [reps]   (metres; res 2 mm)
e.g. the example_data grid 118 x 51 x 133: 0.236 0.102 0.266; the 1000-cell skip limit of eval.py:200: 2.0 0.4 0.4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cppf2_amd import ops
from cppf2_amd.pipeline import VotingPipeline

mode = int(sys.argv[1])
ext = np.array([float(v) for v in sys.argv[2:5]], dtype=np.float32)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
N, T, R, res = 4096, 20000, 180, 2e-3
dev = torch.device("cuda")
rng = np.random.default_rng(0)
cells = int(np.prod(np.floor(ext / res).astype(np.int64) + 2))
pipe = VotingPipeline([N] * B, [T] * B, num_rots=R, res=res, vote_mode=mode, cells_cap=int(cells * 1.1) + 4096)
# points: uniform in the box (the vote kernel sees points, pair indices and the two vote parameters per pair); vote circles of
# radius <= a third of the smallest extent around centres inside the box, so that nearly every vote lands in the grid
pts = torch.from_numpy((rng.random((B * N, 3), dtype=np.float32) - 0.5) * ext + np.float32([0, 0, 0.8])).to(dev)
idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
rad = float(ext.min()) / 3
pipe.tr[:, 0] = torch.from_numpy((rng.random(B * T, dtype=np.float32) - 0.5) * rad).to(dev)
pipe.tr[:, 1] = torch.from_numpy(rng.random(B * T, dtype=np.float32) * rad + 4 * res).to(dev)
pipe.vote_center(pts, idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pipe.vote_center(pts, idx)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
g = pipe.grids.cpu().numpy().view(np.int32).reshape(B, 8)
print("mode %d extent %s B %d: grid %s = %d cells/scene, %.3f ms per vote_center (bounds + zero + votes + argmax), peak %d, %.3g votes/s"
      % (mode, ext.tolist(), B, g[0, 3:6].tolist(), int(g[0, 6]), ms, int(pipe.peak.max()), B * T * R / (ms / 1e3)))
