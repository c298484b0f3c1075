"""Per-workgroup checksums of the rotation vote's intermediate values, solo and beside an MLP kernel of another stream (ROT_TRACE=2)."""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppf2_amd import _lib
_lib.LIB_PATH = os.path.abspath("scratch/rotdbg/lib_trace4.so")
import torch
import bench
from cppf2_amd import models, ops
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.Step(args, 0, 1, dev)
st.run(); torch.cuda.synchronize()
pipe = st.pipe
L = _lib.load()
L.cppf_debug_set_rot_trace.restype = C.c_int; L.cppf_debug_set_rot_trace.argtypes = [C.c_void_p, C.c_longlong]
trace = torch.zeros((64, 4096, 4), dtype=torch.int32, device=dev)
assert L.cppf_debug_set_rot_trace(trace.data_ptr(), 0) == 0
idx = ops.sample_tuples(4096, 20000, 5, 0, tuple(range(64)), dev)
side = torch.cuda.Stream()
g = torch.Generator(device="cpu").manual_seed(1)
w1 = (torch.randn(256, 256, generator=g) / 16).to(dev); w2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
wq = models.pack_split(w1, None, w2, 256); b1 = torch.zeros(256, device=dev)
x256 = torch.randn(400000, 256, device=dev)
trace.zero_(); pipe.rot_bins(st.pts, idx); torch.cuda.synchronize()
ref_counts = pipe.counts.clone(); ref = trace.clone()
trace.zero_(); pipe.rot_bins(st.pts, idx); torch.cuda.synchronize()
print("solo again: checksums equal", torch.equal(trace, ref))
for rep in range(6):
    trace.zero_()
    with torch.cuda.stream(side):
        ops.reslayer_split(x256, wq, b1, None, 256)
    pipe.rot_bins(st.pts, idx)
    torch.cuda.synchronize()
    d = (trace != ref)
    print("rep", rep, "counts differ:", not torch.equal(pipe.counts, ref_counts), "| workgroups whose checksums differ:", int(d.any(-1).sum()),
          "| xyz", int(d[..., 0].sum()), "row+tn", int(d[..., 1].sum()), "u+nn", int(d[..., 2].sum()), "votes+dots", int(d[..., 3].sum()), flush=True)
    for b, blk in d.any(-1).nonzero()[:4].tolist():
        print("   scene %d workgroup %d: solo %s beside %s" % (b, blk, ref[b, blk].tolist(), trace[b, blk].tolist()))
