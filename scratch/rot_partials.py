"""Unmodified rotation-vote kernel (LDS floor off): per-workgroup partial sums solo vs beside an MLP kernel of another stream."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppf2_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("CPPF_PROBE_LIB", "cppf2_amd/libcppf_hip.so"))
import numpy as np, torch
import bench
from cppf2_amd import models, ops
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.Step(args, 0, 1, dev)
st.run(); torch.cuda.synchronize()
pipe = st.pipe
B, R, S, bmm = pipe.B, pipe.R, pipe.S, pipe.bmm
rows = max(pipe.max_kept, 1) * R
nchunks = (rows + bmm - 1) // bmm
target = 160 * R
sub = 1 if target >= bmm else (bmm + target - 1) // target
rpb = (bmm + sub - 1) // sub
nblk = nchunks * sub
print("max_kept", pipe.max_kept, "rows", rows, "bmm", bmm, "nchunks", nchunks, "sub", sub, "rpb", rpb, "nblk", nblk, "S", S)
def partials():
    return pipe.ws[: B * nblk * 2 * S * 8].view(torch.float64).view(B, nblk, 2, S).clone()
idx = ops.sample_tuples(4096, 20000, 5, 0, tuple(range(64)), dev)
side = torch.cuda.Stream()
g = torch.Generator(device="cpu").manual_seed(1)
w1 = (torch.randn(256, 256, generator=g) / 16).to(dev); w2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
wq = models.pack_split(w1, None, w2, 256); b1 = torch.zeros(256, device=dev)
x256 = torch.randn(400000, 256, device=dev)
pipe.rot_bins(st.pts, idx); torch.cuda.synchronize()
ref = partials(); ref_counts = pipe.counts.clone()
kc = pipe.kept_count.cpu().numpy(); row0 = pipe.kept_row0.cpu().numpy(); wt = pipe.kept_wt.cpu().numpy(); toff = pipe.tup_off.cpu().numpy()
for rep in range(6):
    with torch.cuda.stream(side):
        ops.reslayer_split(x256, wq, b1, None, 256)
    pipe.rot_bins(st.pts, idx)
    torch.cuda.synchronize()
    cur = partials()
    d = cur - ref
    wg = (d != 0).any(-1).any(-1).nonzero()
    print("rep", rep, "counts differ:", not torch.equal(pipe.counts, ref_counts), "| workgroups with different partials:", wg.shape[0], flush=True)
    for b, blk in wg[:5].tolist():
        chunk, sb = blk // sub, blk % sub
        lo = chunk * bmm + sb * rpb; hi = min(lo + rpb, (chunk + 1) * bmm)
        t0 = toff[b]
        js = [j for j in range(kc[b]) if row0[t0 + j] >= 0 and row0[t0 + j] < hi and row0[t0 + j] + R > lo]
        inv = {j: 1.0 / wt[t0 + j] for j in js}
        for a in range(2):
            nz = d[b, blk, a].nonzero().flatten().tolist()
            if not nz: continue
            print("   scene %d workgroup %d (rows %d..%d, %d pairs) axis %d: %d bins differ, sum of differences %.6g" % (b, blk, lo, hi, len(js), a, len(nz), float(d[b, blk, a].sum())))
            for s_ in nz[:8]:
                dv = float(d[b, blk, a, s_])
                near = min(inv.items(), key=lambda kv: abs(abs(dv) - kv[1])) if inv else (None, 0)
                print("      bin %d: %+.9g  (closest 1/weight: pair %s %.9g)" % (s_, dv, near[0], near[1]))
