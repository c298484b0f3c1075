"""Which intermediate value of the rotation vote differs beside an MLP kernel of another stream?  (probe library with ROT_TRACE)"""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppf2_amd import _lib
_lib.LIB_PATH = os.path.abspath("scratch/rotdbg/lib_trace.so")
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
rows = int(pipe.kept_count.max().item()) * 180
trace = torch.zeros((64, rows, 2, 16), dtype=torch.float32, device=dev)
assert L.cppf_debug_set_rot_trace(trace.data_ptr(), rows) == 0
idx = ops.sample_tuples(4096, 20000, 5, 0, tuple(range(64)), dev)
side = torch.cuda.Stream()
g = torch.Generator(device="cpu").manual_seed(1)
w1 = (torch.randn(256, 256, generator=g) / 16).to(dev); w2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
wq = models.pack_split(w1, None, w2, 256); b1 = torch.zeros(256, device=dev)
x256 = torch.randn(400000, 256, device=dev)
names = ["x", "y", "z", "phi", "ci", "cj", "e.x", "e.y", "nvote", "dsum", "ux", "uy", "uz", "nn", "tn", "inv_wt"]
pipe.rot_bins(st.pts, idx); torch.cuda.synchronize()
ref_counts = pipe.counts.clone(); ref = trace.clone()
for rep in range(4):
    trace.zero_()
    with torch.cuda.stream(side):
        ops.reslayer_split(x256, wq, b1, None, 256)
    pipe.rot_bins(st.pts, idx)
    torch.cuda.synchronize()
    diff = (trace.view(torch.int32) != ref.view(torch.int32))
    rowsd = diff.any(-1).nonzero()
    print("rep", rep, "counts differ:", not torch.equal(pipe.counts, ref_counts), "| trace records that differ:", rowsd.shape[0], "| fields:",
          {names[i]: int(diff[..., i].sum()) for i in range(16) if int(diff[..., i].sum())}, flush=True)
    for r in rowsd[:6].tolist():
        b, row, a = r
        print("  scene %d row %d (pair %d rot %d) axis %d" % (b, row, row // 180, row % 180, a))
        for i in range(16):
            u, v = ref[b, row, a, i], trace[b, row, a, i]
            if u.view(torch.int32) != v.view(torch.int32):
                print("     %-6s solo %.9g (%08x)  beside %.9g (%08x)" % (names[i], float(u), int(u.view(torch.int32)) & 0xffffffff, float(v), int(v.view(torch.int32)) & 0xffffffff))
