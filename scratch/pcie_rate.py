"""PCIe-inclusive rates of the boundary's host-buffer forms (DESIGN.md "Host buffers"; never the bench's `value`).
  1. the bench step with its inputs (the batch's points) coming from pinned host memory every step and its records going back to
     pinned host memory every step: (a) copies enqueued asynchronously on the step's stream, one synchronisation at the end;
     (b) the host waits for every step's records before it submits the next step (the reference's calling pattern);
  2. `shot.compute(pc)` as the reference calls it (src_shot/shot.cpp:45-100: numpy in, numpy out) per cloud, against the
     device-resident batch entry point.
usage: python scratch/pcie_rate.py [--steps 30]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cppf2_amd.benchlib import launch
from cppf2_amd.benchlib.workloads import Step, Cfg
from cppf2_amd import shot as shotmod, ops, synth

args = launch.parse([a for a in sys.argv[1:]])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
step = Step(args, 0, 1, dev)
step.run(); torch.cuda.synchronize()
K = args.steps
out = {}


def loop(body):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


out["resident_ms"] = 1e3 * loop(step.run)
host_pts = step.pts.cpu().pin_memory()
host_rec = torch.empty((step.B, 160), dtype=torch.uint8).pin_memory()


def async_body():
    step.pts.copy_(host_pts, non_blocking=True)
    step.run()
    host_rec.copy_(step.all_records, non_blocking=True)


def sync_body():
    async_body()
    torch.cuda.current_stream().synchronize()


out["host_buffers_async_ms"] = 1e3 * loop(async_body)
out["host_buffers_sync_each_step_ms"] = 1e3 * loop(sync_body)
pageable = step.pts.cpu()


def pageable_body():
    step.pts.copy_(pageable)
    step.run()
    _ = step.all_records.cpu()


out["pageable_host_buffers_sync_each_step_ms"] = 1e3 * loop(pageable_body)
for k in list(out):
    out[k.replace("_ms", "_scenes_per_s")] = step.B / (out[k] / 1e3)
out["bytes_in_per_step"] = int(host_pts.numel() * 4)
out["bytes_out_per_step"] = int(host_rec.numel())

# 2. shot.compute, numpy in / numpy out, one cloud per call (the reference's calling pattern, eval.py:210)
pcs = [synth.make_scene(args.seed, b, args.points)["pc"] for b in range(16)]
shotmod.compute(pcs[0], Cfg.res * 10, Cfg.res * 10)
t0 = time.perf_counter()
for i in range(64):
    s, n = shotmod.compute(pcs[i % 16], Cfg.res * 10, Cfg.res * 10)
out["shot_compute_numpy_ms_per_cloud"] = 1e3 * (time.perf_counter() - t0) / 64
out["shot_compute_bytes_out_per_cloud"] = int(s.nbytes + n.nbytes)
pts = torch.from_numpy(np.concatenate(pcs)).to(dev)
off = ops._offsets([args.points] * 16, dev)
shotmod.compute_device(pts, off, Cfg.res * 10, Cfg.res * 10); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    shotmod.compute_device(pts, off, Cfg.res * 10, Cfg.res * 10)
torch.cuda.synchronize()
out["shot_compute_device_batch16_ms_per_cloud"] = 1e3 * (time.perf_counter() - t0) / 20 / 16
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))
