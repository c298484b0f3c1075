"""rot_bins at the bench workload: fused two-axis lookup-table kernel vs the single-axis entry point vs the exhaustive
sweep (counts must be identical), and ms per launch (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth
from cppf2_amd.pipeline import VotingPipeline

B, N, T, R = int(os.environ.get("B", 64)), 4096, 20000, 180
reps = int(os.environ.get("REPS", 20))
dev = torch.device("cuda")
scenes = [synth.make_scene(0, b, N) for b in range(B)]
pts = torch.from_numpy(np.concatenate([s["pc"] for s in scenes])).to(dev)
pipe = VotingPipeline([N] * B, [T] * B, num_rots=R)
idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
canon = torch.from_numpy(np.concatenate([s["pc_canon"] for s in scenes])).to(dev)
base = (torch.arange(B, device=dev, dtype=torch.int64) * N).repeat_interleave(T)
coords = canon[(idx[:, :2].long() + base[:, None]).reshape(-1)].reshape(B * T, 6)
pos = (coords.clamp(-0.5, 0.5) + 0.5) * 31.0
kbin = torch.arange(32, device=dev, dtype=torch.float32)
logits = (-0.5 * ((kbin[None, None, :] - pos[..., None]) / 0.6) ** 2).contiguous()
u = ops.philox_uniform(T, 6, 0, 1, tuple(range(B)))
pipe.decode(pts, idx, logits, u)
pipe.vote_center(pts, idx)
pipe.backvote(pts, idx)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


pipe.rot_bins(pts, idx)
fused = pipe.counts.cpu().numpy().copy(); ftop = pipe.top_idx.cpu().numpy().copy()
pipe.counts.zero_()
pipe.rot_bins_single(pts, idx, 0); pipe.rot_bins_single(pts, idx, 1)
single = pipe.counts.cpu().numpy().copy()
print("fused == single:", np.array_equal(fused, single))
if os.environ.get("DENSE", "1") == "1" and B <= 8:
    pipe.rot_bins(pts, idx, use_lut=False)
    dense = pipe.counts.cpu().numpy().copy()
    print("fused == dense:", np.array_equal(fused, dense), "max diff", np.abs(fused - dense).max(), "top equal", np.array_equal(ftop, pipe.top_idx.cpu().numpy()))
print("B", B, "fused ms %.3f" % timeit(lambda: pipe.rot_bins(pts, idx)),
      "single x2 ms %.3f" % timeit(lambda: (pipe.rot_bins_single(pts, idx, 0), pipe.rot_bins_single(pts, idx, 1))))
print("kept", pipe.kept_count[:4].tolist(), "top", ftop[:, :4].tolist())
