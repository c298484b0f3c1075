"""Does running consecutive steps on two HIP streams (independent scene batches: step i + 1's descriptor / voting kernels beside
step i's MLP) raise the throughput?  usage: python scratch/pipeline_probe.py [depth]"""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
steps = [bench.Step(args, 0, 1, dev) for _ in range(depth)]
streams = [torch.cuda.Stream() for _ in range(depth)]
for s, st in zip(steps, streams):
    with torch.cuda.stream(st):
        s.run(); s.run()
torch.cuda.synchronize()
for d in sorted({1, depth}):
    K = 40
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % d]):
            steps[i % d].run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("depth %d: %.3f ms/step, %.0f scenes/s" % (d, 1e3 * dt / K, 64 * K / dt), flush=True)
a = steps[0].pipe.results_to_numpy(); b = steps[-1].pipe.results_to_numpy()
print("records equal across pipelines:", a.tobytes() == b.tobytes())
import numpy as np
for f in a.dtype.names:
    if not np.array_equal(a[f], b[f]):
        bad = np.nonzero(np.any((a[f] != b[f]).reshape(64, -1), axis=1))[0]
        print("field", f, "differs in scenes", bad[:10], "e.g.", a[f][bad[0]], b[f][bad[0]])
# sequential re-runs of each pipeline alone
for s, st in zip(steps, streams):
    with torch.cuda.stream(st):
        s.run()
    torch.cuda.synchronize()
a2 = steps[0].pipe.results_to_numpy(); b2 = steps[-1].pipe.results_to_numpy()
print("sequential: pipelines equal", a2.tobytes() == b2.tobytes(), "| pipeline 0 concurrent == sequential", a.tobytes() == a2.tobytes(), "| last", b.tobytes() == b2.tobytes())
