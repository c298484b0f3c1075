"""Stress test of the split/merge path of the persistent vote kernel against the global-atomic path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth
from cppf2_amd.pipeline import VotingPipeline
dev = torch.device("cuda")
rng = np.random.RandomState(0)
bad = 0
for it in range(40):
    B = int(rng.randint(1, 20))
    Ns = [int(rng.randint(300, 4097)) for _ in range(B)]
    Ts = [int(rng.randint(1500, 20001)) for _ in range(B)]
    scs = [synth.make_scene(100 + it, b, n) for b, n in enumerate(Ns)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = torch.cat([ops.sample_tuples(n, t, 5, it, (b,)) for b, (n, t) in enumerate(zip(Ns, Ts))])
    off = np.cumsum([0] + Ts)
    lg = torch.cat([torch.from_numpy(synth.teacher_logits(s["pc_canon"], idx[off[b]:off[b + 1]].cpu().numpy(), 32)) for b, s in enumerate(scs)]).to(dev)
    u = torch.cat([ops.philox_uniform(t, 6, it, 1, (b,)) for b, t in enumerate(Ts)])
    R = int(rng.choice([36, 90, 180]))
    ref = None
    for mode in (2, 0, 0, 0, 0, 0):
        pipe = VotingPipeline(Ns, Ts, num_rots=R, vote_mode=mode)
        pipe.decode(pts, idx, lg, u)
        grid = torch.zeros(B * pipe.cells_cap, dtype=torch.int32, device=dev)
        goff = torch.arange(B, dtype=torch.int64, device=dev) * pipe.cells_cap
        pipe.vote_center(pts, idx, grid=grid, grid_off=goff)
        out = (pipe.argmax.cpu().numpy().copy(), pipe.peak.cpu().numpy().copy(), int(grid.long().sum()), grid.cpu().numpy())
        if ref is None:
            ref = out
        else:
            ok = np.array_equal(ref[0], out[0]) and np.array_equal(ref[1], out[1]) and ref[2] == out[2] and np.array_equal(ref[3], out[3])
            bad += 0 if ok else 1
print("iterations 40 x 5 persistent runs, mismatches:", bad)
