"""Soak of the ensemble's two-stream mode (bench.EnsembleStep.run_two_streams = the order eval.run_ensemble uses): the selected
records, both passes' records and the losses after every step against the single-stream ones.  usage: python scratch/ensemble_soak.py [steps]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.EnsembleStep(args, 0, 1, dev)
st.run(); torch.cuda.synchronize()
want = (st.pipe.selected.clone(), st.pipe.result_slots.clone(), st.pipe.losses.clone())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for i in range(n):
    st.run_two_streams(streams)
    torch.cuda.synchronize()
    ok = torch.equal(st.pipe.selected, want[0]) and torch.equal(st.pipe.result_slots, want[1]) and torch.equal(st.pipe.losses, want[2])
    bad += int(not ok)
print("ensemble two-stream soak: %d steps (64 instances, both models), %d differ from the single-stream records / losses" % (n, bad))
