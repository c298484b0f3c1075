"""Whole bench step (sampler .. pose record) at small batch sizes: eager launches vs one HIP-graph replay.
usage: python scratch/step_graph.py [B ...]"""
import sys, os, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda")
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    args = types.SimpleNamespace(scenes_per_gpu=B, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
    st = bench.Step(args, 0, 1, dev)
    for _ in range(3):
        st.run()
    torch.cuda.synchronize()
    want = st.pipe.results.clone()

    def timeit(fn, n=200):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    eager = timeit(st.run)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        st.run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st.run()
    st.pipe.results.zero_()
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(st.pipe.results, want)
    graph = timeit(g.replay)
    print("B = %d: eager %.3f ms, HIP-graph replay %.3f ms per step (records identical: %s)" % (B, eager, graph, same))
