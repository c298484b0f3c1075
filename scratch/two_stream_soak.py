"""Soak of the two-stream batch mode: whole bench steps (different scene batches) alternating between two HIP streams -- claimed row
blocks, one CU per shader engine reserved, like bench.py's headline loop -- with the records, bins and rotation counts of BOTH
pipelines compared with their single-stream ones after every pair of steps.  usage: python scratch/two_stream_soak.py [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd.benchlib import launch
from cppf2_amd.benchlib.workloads import Step
from cppf2_amd import ops
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = launch.parse([])
dev = torch.device("cuda")
steps = [Step(args, 0, 1, dev), Step(args, 0, 1, dev, scene_shift=args.scenes_per_gpu)]
want = []
for s in steps:
    s.run(); torch.cuda.synchronize()
    want.append((s.pipe.results.clone(), s.pipe.bins.clone(), s.pipe.counts.clone()))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
ops.mlp_reserve_cus(ops.batch_mode_reserved_cus(dev))
for p in range(pairs):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            steps[i].run()
    if p % 4 == 3:                       # (four pairs enqueued back to back: the streams really run ahead of each other)
        torch.cuda.synchronize()
        for s, w in zip(steps, want):
            ok = torch.equal(s.pipe.results, w[0]) and torch.equal(s.pipe.bins, w[1]) and torch.equal(s.pipe.counts, w[2])
            bad += int(not ok)
ops.mlp_reserve_cus(0)
torch.cuda.synchronize()
print("two-stream soak: %d pairs of overlapped steps (64 scenes each, claimed row blocks %s, %d CUs reserved), %d of %d checks differ from the "
      "single-stream records / bins / rotation counts; counters zero: %s" % (pairs, ops.DYNAMIC_BLOCKS, ops.batch_mode_reserved_cus(dev), bad, 2 * (pairs // 4),
                                                                          all(int(b.abs().sum()) == 0 for b in ops._SCHED.values())))
