"""Micro-benchmark of single stages with the bench workload's inputs (not part of the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd import ops, synth
from cppf2_amd.pipeline import VotingPipeline

B, N, T, R = int(os.environ.get("B", 64)), 4096, 20000, 180
stage = sys.argv[1] if len(sys.argv) > 1 else "vote_center"
reps = int(os.environ.get("REPS", 10))
mode = int(os.environ.get("MODE", 0))
dev = torch.device("cuda")
mk = synth.make_scene_voxel2mm if os.environ.get("CLOUD") == "voxel2mm" else synth.make_scene
scenes = [mk(0, b, N) for b in range(B)]
SCALE = float(os.environ.get("SCALE", 1.0))          # object extent x SCALE at the same 2 mm cells (laptop-sized grids: SCALE=2.5)
pts = torch.from_numpy(np.concatenate([(s["pc"] - s["pc"].mean(0)) * np.float32(SCALE) + s["pc"].mean(0) for s in scenes]).astype(np.float32)).to(dev)
pipe = VotingPipeline([N] * B, [T] * B, num_rots=R, vote_mode=mode)
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
pipe.rot_bins(pts, idx)
torch.cuda.synchronize()
pipe.vote_center(pts, idx, phase=1)
fn = {"vote_center": lambda: pipe.vote_center(pts, idx), "vote_only": lambda: pipe.vote_center(pts, idx, phase=2), "backvote": lambda: pipe.backvote(pts, idx),
      "rot_bins": lambda: pipe.rot_bins(pts, idx), "decode": lambda: pipe.decode(pts, idx, logits, u)}[stage]
fn(); torch.cuda.synchronize()
if stage == "vote_only":
    # the work list belongs to phase 1: run it before every timed phase 2
    evs = []
    for _ in range(reps):
        pipe.vote_center(pts, idx, phase=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / reps
else:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
res = pipe.grids.cpu().numpy().view(np.int32).reshape(B, 8)
print(stage, "mode", mode, "B", B, "ms/launch %.3f" % ms, "mean cells", res[:, 6].mean(), "argmax0", int(pipe.argmax[0]))

if stage in ("vote_center", "vote_only") and (mode & 0xff) in (0, 1) and os.environ.get("DIAG", "0") == "1":
    import ctypes
    smp = max((pipe.cells_cap + 36864 - 1) // 36864, 32)
    raw = pipe.ws[:B * smp * 16].cpu().numpy().view(np.uint32).reshape(B, smp, 4)
    nsl = (res[:, 6] + 36863) // 36864
    durs = np.concatenate([raw[b, :nsl[b], 3] for b in range(B)]).astype(np.float64) / 100.0   # us
    peaks = np.concatenate([raw[b, :nsl[b], 2] for b in range(B)])
    print("WGs", len(durs), "dur us: mean %.0f  p50 %.0f  p90 %.0f  max %.0f  sum/256 %.0f" % (durs.mean(), np.median(durs), np.percentile(durs, 90), durs.max(), durs.sum() / 256))
    b0 = raw[0, :nsl[0], 3] / 100.0
    print("scene0 per-slab us:", np.round(b0).astype(int).tolist())
    # list-scheduling replay of the dispatch: WG id = rank * B + bx -> XCD id % 8, 32 CUs per XCD, greedy
    import heapq
    smax = int(nsl.max())
    order = []
    for rank in range(smax):
        for bx in range(B):
            b = (bx + rank) % B
            if rank < nsl[b]:
                mid = nsl[b] >> 1; dd = (rank + 1) >> 1
                s = mid - dd if (rank & 1) else mid + dd
                order.append((rank * B + bx, raw[b, s, 3] / 100.0))
            else:
                order.append((rank * B + bx, 0.5))
    xcd = [[0.0] * 32 for _ in range(8)]
    for h in xcd: heapq.heapify(h)
    for wid, d in order:
        h = xcd[wid % 8]
        t = heapq.heappop(h); heapq.heappush(h, t + d)
    print("replay makespan per XCD:", [int(max(h)) for h in xcd], " global greedy:", end=" ")
    h = [0.0] * 256; heapq.heapify(h)
    for wid, d in order:
        t = heapq.heappop(h); heapq.heappush(h, t + d)
    print(int(max(h)), " LPT:", end=" ")
    h = [0.0] * 256; heapq.heapify(h)
    for d in sorted([d for _, d in order], reverse=True):
        t = heapq.heappop(h); heapq.heappush(h, t + d)
    print(int(max(h)))
