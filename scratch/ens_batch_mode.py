"""Ensemble workload: the two model passes of ONE batch on two streams (eval.run_ensemble's mode) against whole ensemble steps of
DIFFERENT batches alternating between two streams (pipeline.BatchMode).  usage: python scratch/ens_batch_mode.py [--steps 30]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd.benchlib import launch
from cppf2_amd.benchlib.workloads import EnsembleStep
from cppf2_amd.pipeline import BatchMode
from cppf2_amd import ops
args = launch.parse(sys.argv[1:])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
a = EnsembleStep(args, 0, 1, dev)
b = EnsembleStep(args, 0, 1, dev, scene_shift=args.scenes_per_gpu)
K = args.steps
refs = []
for s in (a, b):
    s.run(); torch.cuda.synchronize(); refs.append(s.pipe.selected.clone())


def loop(body, n=K):
    for _ in range(3): body()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): body()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


single = loop(a.run)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ops.mlp_reserve_cus(ops.batch_mode_reserved_cus(dev))
fork = loop(lambda: a.run_two_streams(streams))
ops.mlp_reserve_cus(0)
a.run(); torch.cuda.synchronize()
out = {}
for r in (0, None):
    with BatchMode([a, b], device=dev, reserve_cus=r) as mode:
        def body():
            with mode.next() as s:
                s.run()
        t = loop(body)
    torch.cuda.synchronize()
    out[r] = t
    print("batch mode reserve", r, "same records:", torch.equal(a.pipe.selected, refs[0]), torch.equal(b.pipe.selected, refs[1]))
B = args.scenes_per_gpu
print("instances/s: one stream %.0f | two passes on two streams %.0f | batch mode %.0f (reserve 0) %.0f (one CU per SE)" % (B / single, B / fork, B / out[0], B / out[None]))
