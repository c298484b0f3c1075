"""Soak of the two-stream mode: whole bench steps alternating between two HIP streams, records of BOTH pipelines compared with the
single-stream records after every pair of steps.  usage: python scratch/two_stream_soak.py [pairs]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
steps = [bench.Step(args, 0, 1, dev) for _ in range(2)]
steps[0].run(); torch.cuda.synchronize()
want = steps[0].pipe.results.clone(); want_bins = steps[0].pipe.bins.clone(); want_counts = steps[0].pipe.counts.clone()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for p in range(pairs):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            steps[i].run()
    torch.cuda.synchronize()
    for s in steps:
        ok = torch.equal(s.pipe.results, want) and torch.equal(s.pipe.bins, want_bins) and torch.equal(s.pipe.counts, want_counts)
        bad += int(not ok)
print("two-stream soak: %d pairs of overlapped steps (64 scenes each), %d of %d step results differ from the single-stream ones (records, bins, rotation counts)" % (pairs, bad, 2 * pairs))
