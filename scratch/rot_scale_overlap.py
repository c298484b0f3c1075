"""rot_bins and the kept-pairs scale head at the bench workload: alone, back to back on one stream, and overlapped on
two streams the way bench.py runs them (not part of the product)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.Step(a, 0, 1, dev)
st.run(); st.run(); torch.cuda.synchronize()
pipe, model = st.pipe, st.model
idx = st.ops.sample_tuples(st.N, st.T, 5, 0, tuple(range(st.B)), dev)
feat = torch.randn((st.B * st.T, 256), device=dev)


def scale_chain():
    rows = pipe.kept_rows()
    return pipe.scatter_kept(rows, model.scale_head(feat[rows]), out=st.scales_buf)


def rot():
    pipe.rot_bins(st.pts, idx)


def both_overlapped():
    cur = torch.cuda.current_stream()
    st.side.wait_stream(cur)
    with torch.cuda.stream(st.side):
        scale_chain()
    rot()
    cur.wait_stream(st.side)


def both_overlapped_rot_on_side():
    cur = torch.cuda.current_stream()
    st.side.wait_stream(cur)
    with torch.cuda.stream(st.side):
        rot()
    scale_chain()
    cur.wait_stream(st.side)


def timeit(fn, reps=30):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("rot_bins alone        %.3f ms" % timeit(rot))
print("scale chain alone     %.3f ms" % timeit(scale_chain))
print("back to back          %.3f ms" % timeit(lambda: (rot(), scale_chain())))
print("overlapped (bench)    %.3f ms" % timeit(both_overlapped))
print("overlapped (rot side) %.3f ms" % timeit(both_overlapped_rot_on_side))
