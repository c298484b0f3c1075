"""Does a bench step allocate device memory (hipMalloc / hipFree) in steady state?  python scratch/alloc_probe.py [--materialize-tuples]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types, torch
import bench
args = types.SimpleNamespace(gpus=1, steps=1, warmup=0, scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, cpu_scenes=0,
                             vote_mode=0, eager_scale_head=False, materialize_tuples="--materialize-tuples" in sys.argv)
dev = torch.device("cuda:0")
step = bench.Step(args, 0, 1, dev)
for i in range(6):
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    step.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print("step %d: host %.2f ms, total %.2f ms, segments allocated %d freed %d, reserved %.2f GB, alloc retries %d" % (
        i, 1e3 * (t1 - t0), 1e3 * (t2 - t0), s1["segment.all.allocated"] - s0["segment.all.allocated"],
        s1["segment.all.freed"] - s0["segment.all.freed"], s1["reserved_bytes.all.current"] / 1e9, s1["num_alloc_retries"]), flush=True)
step.prepare_events()
for timed in (None, 0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        step.run(timed=timed)
    torch.cuda.synchronize()
    print("back to back, timed=%s: %.2f ms/step" % (timed, 1e2 * (time.perf_counter() - t0)), flush=True)
