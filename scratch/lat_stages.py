"""Per-stage latency of the post-MLP path at small batch sizes (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth
from cppf2_amd.pipeline import VotingPipeline
dev = torch.device("cuda")
for B in (1, 4, 8, 16):
    N, T, R = 4096, 20000, 180
    scenes = [synth.make_scene(0, b, N) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scenes])).to(dev)
    idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
    lg = torch.from_numpy(np.concatenate([synth.teacher_logits(s["pc_canon"], idx[b*T:(b+1)*T].cpu().numpy(), 32) for b, s in enumerate(scenes)])).to(dev)
    u = ops.philox_uniform(T, 6, 0, 1, tuple(range(B)))
    pipe = VotingPipeline([N] * B, [T] * B, num_rots=R)
    stages = [("decode", lambda: pipe.decode(pts, idx, lg, u)), ("vote_center", lambda: pipe.vote_center(pts, idx)),
              ("backvote", lambda: pipe.backvote(pts, idx)), ("rot_bins", lambda: pipe.rot_bins(pts, idx)),
              ("assemble", lambda: pipe.assemble())]
    for _, f in stages: f()
    torch.cuda.synchronize()
    out = []
    for name, f in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        out.append("%s %.3f" % (name, e0.elapsed_time(e1) / 20))
    print("B=%d" % B, "  ".join(out), flush=True)
