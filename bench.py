#!/usr/bin/env python
"""Throughput benchmark of the CPPF++ voting hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                      # no launcher environment: starts its 8 rank processes itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one pass of the whole path over a batch of synthetic scenes resident in HBM:
  sampler -> normals + SHOT352 -> shot_encoder (MLP) -> tuple encode (pair features; the descriptor gather happens inside the
  first MLP kernel) -> tuple MLP (3 launches: gathered 360 -> 128 chain | 128 -> 256 + the logit head's 256-wide layers, tuple
  features tapped | 256 -> 192 + bin draw) -> vote parameters -> centre vote + argmax -> back-vote filter -> both rotation votes
  (one kernel) -> scale head on the kept pairs (MLP) -> pose assembly -> one RCCL all_gather of the 160-byte scene records
  (N > 1).  Every launch of a step is a kernel of libcppf_hip.so (profiles/r5_step_trace.txt).
Workload = BASELINE.json configs[1]: SHOT model, 4096 points x 20 000 tuples per scene, 180 rotations, 720 sphere bins, res 2 mm
('bottle' axes), scenes = seeded synthetic bottle-like clouds (cppf2_amd.synth); weights are random-init (no checkpoints ship
with the reference) plus a fixed teacher logit prior so that votes cluster the way trained weights make them.  Scenes are sharded
over ranks (weak scaling: --scenes-per-gpu each).

This file holds what the contract is about -- the TIMED LOOPS (timed_loop: exactly K steps between two barrier + synchronize
pairs, max over ranks) -- and the CPU-baseline leg (the only code that imports oracle/).  The rest lives in cppf2_amd/benchlib:
launch (command line, rank processes), counters (rocprofv3 passes), workloads (the Step classes), report (the JSON line + roofline
arithmetic), evidence (untimed accuracy checks).

Prints ONE JSON line (rank 0) with the driver's contract fields + `roofline` (the dominant kernel -- the tuple MLP: `frac` =
algorithmic float32 flops / time / bf16 MFMA peak, `frac_executed` = the pipe's utilisation including the operand-splitting
overhead; the longest bandwidth-bound kernel under `roofline.hbm`) + `cpu_baseline` (the oracle timed on the host cores, bounded
sample) + `collective` + untimed accuracy evidence (`mlp_error_vs_f64`, `bin_flip_rate_vs_expf`).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cppf2_amd.benchlib import launch      # noqa: E402  (imports no torch)

launch.use_tuned_gemms()                    # before torch initialises

import numpy as np      # noqa: E402
import torch            # noqa: E402

from cppf2_amd.benchlib import counters, evidence, report                       # noqa: E402
from cppf2_amd.benchlib.workloads import Cfg, DenseStep, EnsembleStep, Step     # noqa: E402


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        return os.cpu_count()


def cpu_baseline(args, step, res):
    """`cpu_baseline` of the headline: the oracle (NumPy + C SHOT) timed on the host for a bounded sample of the same workload, and
    how the GPU records of those scenes agree with it.  Returns (cpu_baseline, oracle_agreement)."""
    from oracle import pipeline_oracle as PO          # the checker, timed as the CPU baseline (never the product path)
    from cppf2_amd import synth
    from cppf2_amd.metrics import rt_degree_cm
    weights = {k: v.detach().cpu().numpy() for k, v in step.model.state_dict().items()}
    trig = (step.pipe.cs.cpu().numpy(), step.pipe.sn.cpu().numpy())
    t0 = time.perf_counter()
    outs = []
    for b in range(args.cpu_scenes):
        sc = step.scenes[b]
        outs.append(PO.run_scene_full(weights, sc["pc"], args.seed, step.scene0 + b, args.tuples, res=Cfg.res, num_rots=args.rots,
                                      trig=trig, prior_fn=lambda idx, sc=sc: synth.teacher_logits(sc["pc_canon"], idx, 32, 0.6)))
    dt = time.perf_counter() - t0
    cpu = dict(value=args.cpu_scenes / dt, unit="scenes/s", cores=host_cores(), kind="port",
               sample="%d scene(s) of the same workload (first scenes of rank 0's batch), NumPy oracle + C SHOT "
                      "oracle (SHOT single-threaded like PCL, matmuls on all cores), %.1f s" % (args.cpu_scenes, dt),
               threads_per_stage={"shot_descriptor (C oracle, like PCL)": 1, "mlp_matmuls (NumPy -> BLAS)": "all cores (BLAS default)",
                                  "decode / votes / back-vote / rotation bins (NumPy)": 1})

    def rt(R, t):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = R, t
        return m
    errs = [rt_degree_cm(rt(res["R"][b], res["t"][b]), rt(o["R_est"], o["T_est"]), "bottle", clip=True) for b, o in enumerate(outs)]
    agree = dict(scenes=len(outs), match_5deg5cm=float(np.mean([e[0] <= 5 and e[1] <= 5 for e in errs])),
                 max_rot_err_deg=float(max(e[0] for e in errs)), max_shift_cm=float(max(e[1] for e in errs)),
                 centre_argmax_equal=int(sum(int(res["argmax"][b]) == o["argmax"] for b, o in enumerate(outs))),
                 up_bin_equal=int(sum(int(res["up_idx"][b]) == o["up_idx"] for b, o in enumerate(outs))),
                 right_bin_equal=int(sum(int(res["right_idx"][b]) == o["right_idx"] for b, o in enumerate(outs))))
    return cpu, agree


def cpu_baseline_ensemble(args, step, both, pick):
    """The same for --workload ensemble: both models' NumPy forward + run_instance_ensemble per instance."""
    from oracle import pipeline_oracle as PO          # the checker, timed as the CPU baseline (never the product path)
    from oracle import cppf_oracle as O
    from oracle import shot_oracle as S
    from cppf2_amd import synth
    N, T, R = step.N, step.T, args.rots
    wd = {k_: v_.detach().cpu().numpy() for k_, v_ in step.dino.state_dict().items()}
    wsh = {k_: v_.detach().cpu().numpy() for k_, v_ in step.model.state_dict().items()}
    trig = (step.pipe.cs.cpu().numpy(), step.pipe.sn.cpu().numpy())
    n_cpu = min(args.cpu_scenes, 2)
    t0 = time.perf_counter()
    outs = []
    for b in range(n_cpu):
        sc = step.scenes[b]
        idx = O.sample_tuples(args.seed, step.scene0 + b, T, 5, N).astype(np.int64)
        shot_feat, normal, _, _ = S.compute_ex(sc["pc"], Cfg.res * 10, Cfg.res * 10, pcl_arithmetic=True)
        shot_feat, normal = np.nan_to_num(shot_feat, nan=0.0), np.nan_to_num(normal, nan=0.0)
        prior = synth.teacher_logits(sc["pc_canon"], idx, 32, 0.6)
        desc = step.desc[b * N:(b + 1) * N].cpu().numpy()
        per_model = []
        for m, (lg, scl) in enumerate((PO.mlp_dino(wd, sc["pc"], desc, idx), PO.mlp_shot(wsh, sc["pc"], idx, shot_feat, normal))):
            per_model.append(((lg + prior).astype(np.float32), scl, O.philox_uniform(args.seed, step.scene0 + b, 1 + m, T, 6)))
        outs.append(PO.run_instance_ensemble(sc["pc"], idx, per_model, Cfg.up, Cfg.right, Cfg.front, Cfg.res, num_rots=R,
                                             y_only=True, trig=trig))
    dtc = time.perf_counter() - t0
    cpu = dict(value=n_cpu / dtc, unit="instances/s", cores=host_cores(), kind="port",
               sample="%d instance(s) of the same workload, both models (NumPy oracle: mlp_dino + mlp_shot + run_instance_ensemble; "
                      "C SHOT oracle), %.1f s" % (n_cpu, dtc),
               threads_per_stage={"shot_descriptor": 1, "mlp_matmuls": "BLAS default (all cores)", "votes_and_bins": 1})
    agree = dict(instances=n_cpu,
                 pick_equal=int(sum(int(pick[b]) == o["pick"] for b, o in enumerate(outs))),
                 centre_argmax_equal=[int(sum(int(both[m]["argmax"][b]) == o["models"][m]["argmax"] for b, o in enumerate(outs))) for m in (0, 1)],
                 up_bin_equal=[int(sum(int(both[m]["up_idx"][b]) == o["models"][m]["up_idx"] for b, o in enumerate(outs))) for m in (0, 1)],
                 max_abs_loss_difference=float(max(abs(float(step.pipe.losses[m][b]) - o["models"][m]["loss"])
                                                   for b, o in enumerate(outs) for m in (0, 1))))
    return cpu, agree


class Timer:
    """The contract's timing protocol, shared by every loop of the file: EXACTLY k steps between two (barrier + synchronize) pairs,
    the wall-clock seconds reduced with MAX over ranks."""

    def __init__(self, dev):
        self.dev = dev

    def sync(self):
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def reduce(self, seconds, op=None):
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        if torch.distributed.is_initialized():
            if torch.distributed.get_backend() == "gloo":
                t = t.cpu()
            torch.distributed.all_reduce(t, op=op or torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def loop(self, k, body):
        """body(i) runs step i (and may return its stage events).  Returns (max-over-ranks seconds, this rank's seconds, events)."""
        self.sync()
        t0 = time.perf_counter()
        evs = []
        for i in range(k):
            ev = body(i)
            if ev is not None:
                evs.append(ev)
        self.sync()
        mine = time.perf_counter() - t0
        return self.reduce(mine), mine, evs

    def per_rank(self, mine, steps):
        """ms per step of the slowest and the fastest rank (the spread a first multi-GPU run wants to see)."""
        if not torch.distributed.is_initialized():
            return {"min": 1e3 * mine / steps, "max": 1e3 * mine / steps}
        return {"min": 1e3 * self.reduce(mine, torch.distributed.ReduceOp.MIN) / steps, "max": 1e3 * self.reduce(mine) / steps}


def sampled_steps(k, slots):
    """Which of k steps record their stage boundaries, and into which prepared event slot: up to `slots`, spread evenly."""
    n_s = min(k, slots)
    return {int(round((j + 0.5) * k / n_s - 0.5)): j for j in range(n_s)} if n_s else {}


def completion_intervals(ends, group=1):
    """ms per step between step completions.  group = number of streams: concurrent steps run side by side and complete
    within a millisecond of each other, so the figure is the time from completion i to completion i + group, divided by group."""
    if len(ends) < group + 2:
        return None
    t = sorted(ends[0].elapsed_time(e) for e in ends)
    d = sorted((b_ - a_) / group for a_, b_ in zip(t[:-group], t[group:]))
    return {"min": round(d[0], 4), "median": round(d[len(d) // 2], 4), "max": round(d[-1], 4), "n": len(d),
            "p05": round(d[int(0.05 * len(d))], 4), "p95": round(d[int(0.95 * len(d))], 4), "steps_per_interval": group}


def run_other_workload(args, rank, world, dev, backend, timer):
    """--workload ensemble | dense64k: single-stream loop with stage events; ensemble also its two-stream batch mode."""
    step = EnsembleStep(args, rank, world, dev) if args.workload == "ensemble" else DenseStep(args, rank, world, dev)
    step.prepare_events()
    step.run()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step.run()
    sampled = sampled_steps(args.steps, Step.EVENT_SLOTS)
    dt1, mine, evs = timer.loop(args.steps, lambda i: step.run(timed=sampled.get(i)))      # single stream: stage events
    step.two = None
    dt = dt1
    if args.workload == "ensemble" and not args.single_stream:
        # the product's batch mode (eval.run_ensemble): the two model passes on two HIP streams with twin pipelines
        torch.cuda.synchronize()
        ref_sel = step.pipe.selected.clone()
        ref_slots = step.pipe.result_slots.clone()
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        # like eval.run_ensemble: one CU per shader engine left to the other pass' kernels (with the row blocks claimed from a
        # counter the two passes' wide launches share the chip gracefully: + 1.3 %; with fixed shares it lost 3.5 %, measurements 11.7)
        if args.mlp_reserve_cus is None:
            args.mlp_reserve_cus = step.ops.batch_mode_reserved_cus(dev)
        step.ops.mlp_reserve_cus(args.mlp_reserve_cus)
        for _ in range(max(2, args.warmup)):
            step.run_two_streams(streams)
        dt, mine, _ = timer.loop(args.steps, lambda i: step.run_two_streams(streams))
        step.ops.mlp_reserve_cus(0)
        same = bool(torch.equal(step.pipe.selected, ref_sel) and torch.equal(step.pipe.result_slots, ref_slots))
        step.two = {"streams": 2, "mlp_reserved_cus": args.mlp_reserve_cus, "records_identical_to_single_stream": same, "value_single_stream": step.B * world * args.steps / dt1,
                    "ms_per_step_single_stream": 1e3 * dt1 / args.steps,
                    "note": "the DINO pass and the SHOT pass (descriptors included) on two HIP streams, twin pipelines, one event "
                            "for the DINO scale; per-stage times come from the single-stream loop of the same run"}
    rank_ms = timer.per_rank(mine, args.steps)
    if rank == 0:
        if args.workload == "ensemble":
            line = report.report_ensemble(args, step, dt, evs, world, backend, cpu_fn=cpu_baseline_ensemble, rank_ms=rank_ms)
        else:
            line = report.report_dense(args, step, dt, evs, world, backend, rank_ms=rank_ms)
        print(json.dumps(line))
        return 0 if line.get("ok", True) else 3
    return 0


def main():
    args = launch.parse()
    if args.workload == "dense64k" and "--scenes-per-gpu" not in sys.argv:
        args.scenes_per_gpu = 16
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch.self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torch.distributed.run --nproc-per-node %d), "
              "or run `python bench.py --gpus %d` without a launcher environment and it starts the ranks itself"
              % (args.gpus, world, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if args.counter_child:              # a child of collect_counters: the bare single-stream loop, nothing else
        args.single_stream = args.no_reference_order = args.no_native_arith = args.no_f16x2 = args.no_evidence = args.no_counters = True
        args.no_voxel_density = True
        args.cpu_scenes = 0
    COUNTERS = counters.COUNTERS
    profiled = any(k_.startswith(("ROCP_", "ROCPROF")) for k_ in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled and not args.no_counters:
        # this process is itself being profiled (the tool's library is loaded and has initialised the GPU): no children from here
        args.no_counters = True
        COUNTERS["reason"] = "running under a profiler: the counter passes are started by unprofiled runs only"
    counter_children = 0
    if world == 1 and rank == 0 and not args.no_counters and not launch.cdist_forced():
        # BEFORE this process initialises the GPU: fresh children under rocprofv3, one counter pass each (one rank only: a rank of
        # a multi-GPU run never starts children -- `counter_children` stays 0 and is printed)
        wl = ["--scenes-per-gpu", str(args.scenes_per_gpu), "--points", str(args.points), "--tuples", str(args.tuples), "--rots",
              str(args.rots), "--seed", str(args.seed), "--vote-mode", str(args.vote_mode), "--workload", args.workload, "--cloud", args.cloud]
        wl += ["--mlp-arith", args.mlp_arith] if args.mlp_arith else []
        wl += ["--eager-scale-head"] if args.eager_scale_head else []
        wl += ["--materialize-tuples"] if args.materialize_tuples else []
        wl += ["--separate-encode"] if args.separate_encode else []
        COUNTERS.clear()
        COUNTERS.update(counters.collect_counters(wl))
        counter_children = len(counters.COUNTER_PASSES)
    elif args.no_counters:
        if COUNTERS.get("reason", "not collected") == "not collected":
            COUNTERS["reason"] = "--no-counters"
    else:
        COUNTERS["reason"] = "counter passes run at one rank only (N = 1)"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    counters.set_device_cus(torch.cuda.get_device_properties(dev).multi_processor_count)
    affinity = launch.pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    from cppf2_amd import dist as cdist
    backend = None
    if world > 1 or cdist.force_collective():
        # RCCL over xGMI; CPPF_BENCH_BACKEND=gloo is a dry-run switch for boxes with fewer GPUs than ranks.
        # CPPF_DIST_FORCE_COLLECTIVE=1 makes a one-rank run create the group and issue the all_gather too.
        backend = os.environ.get("CPPF_BENCH_BACKEND", "nccl")
        cdist.init(backend=backend, device=dev if backend == "nccl" else None)
        assert torch.distributed.get_world_size() == world and torch.distributed.get_rank() == rank
    timer = Timer(dev)

    from cppf2_amd import models as _models
    if args.mlp_arith:
        _models.MLP_ARITH = args.mlp_arith
    if args.workload in ("ensemble", "dense64k"):
        assert _models.MLP_ARITH in ("split", "split16"), "--workload %s runs the library's kernels (split arithmetic)" % args.workload
        rc = run_other_workload(args, rank, world, dev, backend, timer)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        sys.exit(rc)

    step = Step(args, rank, world, dev)
    with torch.no_grad():           # the support checks look at the inference mode Step.run() executes in
        report.GATHERED_TUPLES = step.gather
        if report.GATHERED_TUPLES:
            counters.STAGE_KERNEL["encode_tuples"] = "encode_shot_heads_tile_kernel"
            report.ENCODE_FOLDED = (not args.separate_encode) and _models.MLP_ARITH == "split"
            from cppf2_amd.models import decode_supported
            report.FUSED_DRAW = decode_supported(step.model.logit_encoder, torch.empty((1, 256), device=dev))
            if report.FUSED_DRAW:
                counters.STAGE_KERNEL["decode_bins"] = "decode_targets_kernel"
    step.prepare_events()
    step.run()                      # part of the untimed setup: allocator pools of both streams, GEMM solution table,
    torch.cuda.synchronize()        # kernel attributes -- so that even --warmup 0 times steady-state steps
    for _ in range(args.warmup):
        step.run()

    # one completion event per step of the headline loops (recorded on the step's stream right after its last launch): the
    # intervals between consecutive completions are the per-step figures of the line (step_interval_ms)
    end_pool = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def timed_loop(k, sample=True, pair=None, one=None):
        """THE timed region: exactly k steps between two barrier + synchronize pairs; max over ranks of the wall-clock seconds.
        pair = an entered cppf2_amd.pipeline.BatchMode: step i runs on the mode's next stream with that stream's own Step;
        one = the Step to run on the current stream (default: `step`).  Returns (seconds, own seconds, stage events, end events)."""
        sampled = sampled_steps(k, Step.EVENT_SLOTS) if sample else {}
        ends = []
        st = one or step

        def body(i):
            if pair is None:
                ev = st.run(timed=sampled.get(i))
                if sample:
                    end_pool[i].record()
            else:
                with pair.next() as s_:            # cppf2_amd.pipeline.BatchMode: the next stream and its Step
                    ev = s_.run(timed=sampled.get(i))
                    if sample:
                        end_pool[i].record()
            if sample:
                ends.append(end_pool[i])
            return ev
        dt_, mine_, evs_ = timer.loop(k, body)
        return dt_, mine_, evs_, ends

    if os.environ.get("CPPF_BENCH_HOSTTIMES"):
        torch.cuda.synchronize()
        for _ in range(4):              # back to back, like the timed loop
            step.host_times = []
            step.run()
            ht = step.host_times + [("returned", time.perf_counter())]
            print("host enqueue times (ms since step start): " + ", ".join("%s %.3f" % (n_, 1e3 * (t_ - ht[0][1])) for n_, t_ in ht),
                  file=sys.stderr)
        torch.cuda.synchronize()
        step.host_times = None
    # ---- the headline loop.  Default: the product's batch mode -- consecutive steps, i.e. independent scene batches, alternate
    # between TWO HIP streams, each with its own resident state (software pipelining: one batch's descriptor, voting and small MLP
    # kernels run beside the other batch's wide matrix-core kernels).  The pipelines hold DIFFERENT scene batches (the second one the
    # global scenes behind the first's): after the loop each pipeline's records must equal, byte for byte, the records the same
    # Step produced alone on one stream before it (`ok`) -- a buffer aliased between the two streams would change them.  The
    # single-stream loop of rounds 1-3 is timed right after it with the same protocol (value_single_stream); --single-stream makes
    # it the headline.
    two = None
    dt_single = None
    if not args.single_stream:
        ns = max(2, int(args.streams))
        others = [Step(args, rank, world, dev, scene_shift=j * args.scenes_per_gpu * world) for j in range(1, ns)]
        for o_ in others:
            o_.prepare_events()
        from cppf2_amd.pipeline import BatchMode
        all_steps = [step] + others
        refs = []
        for s_ in all_steps:             # every pipeline alone on the current stream: the records of ITS scenes
            s_.run()
            torch.cuda.synchronize()
            refs.append(s_.pipe.results.clone())
        # The library's batch mode (cppf2_amd.pipeline.BatchMode): one stream per Step, and while it is active the persistent MLP
        # launches leave one CU per shader engine to the other stream's kernels (cppf_mlp_reserve_cus; a queue's workgroups are
        # placed round-robin over the engines, so ONE full engine stalls the other stream's whole launch: 31 reserved CUs change
        # nothing, 32 give + 3.5 %, docs/measurements.md 11.7).  --mlp-reserve-cus overrides the count.
        mode = BatchMode(all_steps, device=dev, reserve_cus=args.mlp_reserve_cus)
        args.mlp_reserve_cus = mode.reserve_cus
        streams = mode.streams
        with mode:
            for _ in range(max(2, args.warmup) * ns):
                with mode.next() as s_:
                    s_.run()
            torch.cuda.synchronize()
            dt, mine, evs, ends = timed_loop(args.steps, pair=mode)
        pair = (all_steps, streams)
        intervals = completion_intervals(ends, group=len(streams))
        same = bool(all(torch.equal(s_.pipe.results, r_) for s_, r_ in zip(pair[0], refs)))
        distinct = bool(all(not torch.equal(refs[0], r_) for r_ in refs[1:]))
        dt_single, _, evs_single, ends_single = timed_loop(args.steps)
        intervals_single = completion_intervals(ends_single)
        two = {"streams": ns, "mlp_reserved_cus": args.mlp_reserve_cus, "records_identical_to_single_stream": same, "pipelines_hold_different_scenes": distinct,
               "note": "steps alternate between two HIP streams with double-buffered state; the pipelines hold different scene batches, and "
                       "each one's records are compared byte for byte with what the same pipeline produced alone on one stream in this "
                       "run; per-stage times of the headline are measured on the stage's own stream while the other stream's kernels "
                       "share the chip (per_stage_ms_two_streams; per_stage_ms: the same stages alone)"}
        del others
    else:
        dt, mine, evs, ends = timed_loop(args.steps)
        intervals = completion_intervals(ends)
        intervals_single = None
        evs_single = evs
    rank_ms = timer.per_rank(mine, args.steps)
    # ---- comparison loops (one-rank diagnostics: a multi-GPU run times the headline loop only), same protocol, untimed for the headline
    # the other placement of the scale head (see --eager-scale-head): the reference's forward order
    dt_other = None
    if not args.no_reference_order and world == 1:
        step.eager = not step.eager
        step.run()
        dt_other = timed_loop(args.steps, sample=False)[0]
        step.eager = not step.eager
    # the same step with the tuple MLP on the f32-input matrix cores (the arithmetic of rounds 1-2)
    dt_native = None
    if _models.MLP_ARITH == "split" and not args.no_native_arith and world == 1:
        _models.MLP_ARITH = "native"
        step.run()
        dt_native = timed_loop(args.steps, sample=False)[0]
        _models.MLP_ARITH = "split"
    # the same step with the MLP in f16x2 arithmetic (fp16 operand pairs, three products per K step: half the matrix-core work;
    # error against float64 reported under mlp_error_vs_f64.split_f16x2); not the headline
    dt_f16 = None
    f16_agreement = None
    if _models.MLP_ARITH == "split" and not args.no_f16x2 and world == 1:
        _models.MLP_ARITH = "split16"
        step.run()
        dt_f16 = timed_loop(args.steps, sample=False)[0]
        rec16 = step.pipe.results.clone()
        bins16 = step.pipe.bins.clone()
        _models.MLP_ARITH = "split"
        step.run()                  # the records the rest of the report reads are the headline arithmetic's
        torch.cuda.synchronize()
        # what the f16x2 step produced against the headline arithmetic's, this run's scenes: bins drawn, records byte for byte
        r16, r3_ = step.pipe.results_to_numpy(rec16), step.pipe.results_to_numpy()
        f16_agreement = {"bin_draws": int(bins16.numel()), "bins_differing_from_headline": int((bins16 != step.pipe.bins).sum().item()),
                         "scenes": int(r3_.shape[0]),
                         "scenes_with_equal_argmax_rotation_bins_kept_count": int(sum(
                             all(r16[f][i] == r3_[f][i] for f in ("argmax", "up_idx", "right_idx", "kept"))
                             for i in range(r3_.shape[0]))),
                         "max_abs_scale_difference": float(np.nanmax(np.abs(r16["scale"] - r3_["scale"]))),
                         "max_abs_translation_difference_m": float(np.nanmax(np.abs(r16["t"] - r3_["t"])))}
    # the same path on clouds at the point density real inputs have (--cloud voxel2mm: ~250 neighbours inside the SHOT support)
    voxel = None
    if args.cloud == "synthetic" and not args.no_voxel_density and world == 1:
        vargs = type(args)(**vars(args))
        vstep = Step(vargs, rank, world, dev, cloud="voxel2mm")
        vstep.prepare_events()
        for _ in range(2):
            vstep.run()
        torch.cuda.synchronize()
        kv = min(args.steps, 20)
        dtv, _, evv, _ = timed_loop(kv, one=vstep)
        vms = report.stage_means(evv)
        voxel = {"value": vstep.B * kv / dtv, "unit": "scenes/s", "ms_per_step": 1e3 * dtv / kv, "steps": kv, "streams": 1,
                 "cloud": "voxel2mm (cppf2_amd.synth.make_scene_voxel2mm: one point per 2 mm cell, ~250 neighbours in the 2 cm support)",
                 "shot_stage_ms": round(vms.get("shot_frames", 0.0) + vms.get("shot352", 0.0), 4),
                 "per_stage_ms": {s_: round(vms.get(s_, 0.0), 4) for s_ in Step.STAGES},
                 "pose_5deg5cm_vs_gt": report.pose_ok_vs_gt(vstep.pipe.results_to_numpy(), vstep.scenes) / vstep.B}
        del vstep
        step.run()
        torch.cuda.synchronize()

    if os.environ.get("CPPF_BENCH_PER_STEP") and rank == 0:
        for i_, ev in enumerate(evs):
            row = {n1: round(e0.elapsed_time(e1), 3) for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:])}
            print("step %d: total %.3f  %s" % (i_, ev[0][1].elapsed_time(ev[-1][1]), {k_: v_ for k_, v_ in row.items() if v_ > 0.2}),
                  file=sys.stderr)
    # per-stage HIP-event times (ms per launch, averaged over the sampled steps) on the stream the kernels ran on.  With two streams a
    # stage's event time on its own stream includes the time its kernels wait for the other stream's (two persistent matrix-core
    # kernels do not fit a CU together): the kernels' own durations -- what the roofline divides by -- are the stage times of the
    # single-stream loop of the same run; stage_ms_2s keeps the two-stream stage times for the record.
    stage_ms_2s = report.stage_means(evs)
    stage_ms = report.stage_means(evs_single) if dt_single is not None else stage_ms_2s
    step_times = sorted(ev[0][1].elapsed_time(ev[-1][1]) for ev in evs)

    failed = False
    if rank == 0:
        cpu = None
        if args.cpu_scenes > 0 and world == 1:     # CPU baseline: rank 0 at N=1 only
            cpu = cpu_baseline(args, step, step.pipe.results_to_numpy())
        ev_ = evidence.arithmetic_evidence(step) if world == 1 and not args.no_evidence else {}
        line, problems = report.report_shot(dict(
            args=args, step=step, world=world, backend=backend, dt=dt, dt_single=dt_single, dt_other=dt_other, dt_native=dt_native,
            dt_f16=dt_f16, f16_agreement=f16_agreement, two=two, intervals=intervals, intervals_single=intervals_single,
            stage_ms=stage_ms, stage_ms_2s=stage_ms_2s, step_times=step_times, affinity=affinity, cpu=cpu, evidence=ev_, voxel=voxel,
            rank_ms=rank_ms))
        line["counter_children_started"] = counter_children
        print(json.dumps(line))
        if problems:
            print("bench.py: FAILED self-checks: " + "; ".join(problems), file=sys.stderr)
            failed = True
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
