"""Single-instance latency of the post-MLP path (decode..pose): eager launches vs one HIP-graph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth
from cppf2_amd.pipeline import VotingPipeline
dev = torch.device("cuda")
for B in (1, 8):
    N, T, R = 4096, 20000, 180
    scenes = [synth.make_scene(0, b, N) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scenes])).to(dev)
    idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
    lg = torch.from_numpy(np.concatenate([synth.teacher_logits(s["pc_canon"], idx[b*T:(b+1)*T].cpu().numpy(), 32) for b, s in enumerate(scenes)])).to(dev)
    u = ops.philox_uniform(T, 6, 0, 1, tuple(range(B)))
    pipe = VotingPipeline([N] * B, [T] * B, num_rots=R)
    pipe.vote(pts, idx, lg, u); torch.cuda.synchronize()
    ref = pipe.results.clone()
    def timeit(fn, n=50):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    eager = timeit(lambda: pipe.vote(pts, idx, lg, u))
    replay = pipe.capture(pts, idx, lg, u)
    graph = timeit(replay)
    same = bool(torch.equal(ref, pipe.results))
    print("B=%d  eager %.3f ms  graph %.3f ms  identical results %s" % (B, eager, graph, same))
