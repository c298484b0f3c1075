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
  (N > 1).  Every launch of a step is a kernel of libcppf_hip.so (30 per step, profiles/r3_step_trace.txt); the MLPs are PyTorch
  modules whose ResLayers run as matrix-core kernels (cppf_reslayer_split: float32-equivalent split-bf16 arithmetic;
  --mlp-arith native = library GEMMs on the f32-input matrix cores, also timed in every run as value_f32_input_mfma).
  (--eager-scale-head runs the scale head on every tuple, the order of the reference's forward; its output is read only for the
  kept pairs, eval.py:272.)
Workload = BASELINE.json configs[1]: SHOT model, 4096 points x 20 000 tuples per scene, 180 rotations, 720 sphere
bins, res 2 mm ('bottle' axes), scenes = seeded synthetic bottle-like clouds (cppf2_amd.synth); weights are
random-init (no checkpoints ship with the reference) plus a fixed teacher logit prior so that votes cluster the
way trained weights make them.  Scenes are sharded over ranks (weak scaling: --scenes-per-gpu each).

Prints ONE JSON line (rank 0) with the driver's contract fields + `roofline` (the dominant kernel -- the tuple MLP, matrix-core /
power bound; the longest bandwidth-bound kernel under `roofline.hbm`; HIP-event timed inside the timed region) + `cpu_baseline`
(the oracle timed on the host cores, bounded sample) + `collective` + untimed accuracy evidence measured on the bench's own tuples
(`mlp_error_vs_f64`, `bin_flip_rate_vs_expf`).
"""
import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _use_tuned_gemms():
    """PyTorch TunableOp with the GEMM solutions recorded for this workload's shapes on gfx950
    (cppf2_amd/tunableop/gfx950_bench_shapes.csv, produced by one `PYTORCH_TUNABLEOP_TUNING=1 python bench.py` run;
    +4 % on the fp32 MLP).  No tuning happens at run time; a file recorded for other library versions is ignored by
    PyTorch (validator lines).  Must run before torch initialises; respects an explicit PYTORCH_TUNABLEOP_* setup."""
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return
    src = os.path.join(ROOT, "cppf2_amd", "tunableop", "gfx950_bench_shapes.csv")
    if not os.path.exists(src):
        return
    import atexit
    import shutil
    import tempfile
    # one directory per rank process (self_launch strips these variables from its children, so every rank gets here)
    d = tempfile.mkdtemp(prefix="cppf_tunableop_r%s_" % os.environ.get("RANK", "0"))
    atexit.register(shutil.rmtree, d, True)
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    shutil.copy(src, os.path.join(d, "gemm%d.csv" % dev))          # PyTorch appends the device ordinal to the name
    for k_, v_ in (("PYTORCH_TUNABLEOP_ENABLED", "1"), ("PYTORCH_TUNABLEOP_TUNING", "0"),
                   ("PYTORCH_TUNABLEOP_FILENAME", os.path.join(d, "gemm.csv"))):
        os.environ[k_] = v_
        _SET_HERE.append(k_)


_SET_HERE = []          # environment variables this process set itself (not inherited by the ranks self_launch starts)


_use_tuned_gemms()

import numpy as np      # noqa: E402
import torch            # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
F32_MFMA_PEAK_TFLOPS = 157.0    # f32-input MFMA: 256 CUs x 4 SIMDs x 64 flops/cycle x 2.4 GHz


class Cfg:
    num_more = 3
    res = 2e-3
    up, right, front = [0, 1, 0], [1, 0, 0], [0, 0, 1]     # config/config.yaml:12-14


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes-per-gpu", type=int, default=64)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--tuples", type=int, default=20000)
    ap.add_argument("--rots", type=int, default=180)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-scenes", type=int, default=3, help="scenes of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--vote-mode", type=int, default=0)
    ap.add_argument("--eager-scale-head", action="store_true",
                    help="run the scale head on every tuple like the reference's forward (default: only on the pairs "
                         "that survive the back-vote filter, the only rows eval.py:272 ever reads)")
    ap.add_argument("--no-reference-order", action="store_true",
                    help="skip the second timed loop that measures the other scale-head placement (value_reference_order)")
    ap.add_argument("--mlp-arith", choices=("split", "split16", "native"), default=None,
                    help="arithmetic of the tuple MLP: split = float32 as 3 x bf16 on the bf16 matrix cores (default, exact "
                         "products, cppf_reslayer_split), split16 = float32 as 2 x fp16 (22-23 bits per operand, half the "
                         "matrix-core work, cppf_reslayer_split16), native = f32-input matrix cores (library GEMMs + "
                         "cppf_reslayer128)")
    ap.add_argument("--no-f16x2", action="store_true",
                    help="skip the extra timed loop with the MLP in f16x2 arithmetic (value_f16x2_mfma)")
    ap.add_argument("--materialize-tuples", action="store_true",
                    help="write the [T, 360] tuple rows (cppf_encode_tuples_shot) and let the MLP read them back, instead of "
                         "gathering them inside the first ResLayer's kernel (cppf_reslayer_split_gather)")
    ap.add_argument("--no-native-arith", action="store_true",
                    help="skip the extra timed loop with the MLP on the f32-input matrix cores (value_f32_input_mfma)")
    ap.add_argument("--two-streams", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--single-stream", "--no-two-streams", dest="single_stream", action="store_true",
                    help="time the headline with every step on ONE HIP stream (rounds 1-3).  Default: consecutive steps "
                         "(independent scene batches) alternate between two streams with double-buffered state -- the product's "
                         "batch mode (eval.run_ensemble runs its two model passes the same way); the single-stream figure is "
                         "still measured and printed as value_single_stream")
    ap.add_argument("--no-evidence", action="store_true",
                    help="skip the untimed accuracy evidence (mlp_error_vs_f64, bin_flip_rate_vs_expf)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the headline loop rotates its steps over, each with its own resident state (default 2)")
    ap.add_argument("--no-counters", action="store_true",
                    help="skip the rocprofv3 counter passes (roofline.traffic and the unit-activity fractions are then null)")
    ap.add_argument("--counter-child", action="store_true", help=argparse.SUPPRESS)      # set by collect_counters for its children
    ap.add_argument("--workload", choices=("shot", "ensemble", "dense64k"), default="shot",
                    help="shot (default, the headline): BASELINE configs[1], the SHOT model; ensemble: BASELINE configs[2], the "
                         "reference's real per-instance loop (eval.py:219-372) -- the DINO model AND the SHOT model vote every "
                         "instance, the pose with the smaller alignment loss is kept; value = instances/s; dense64k: BASELINE "
                         "configs[4], 65 536 pairs per scene, float16 feature table, uncertainty-weighted centre votes (extensions "
                         "the reference does not have; --tuples is ignored, --scenes-per-gpu defaults to 16)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-stage table to stderr")
    return ap.parse_args()


class Step:
    """Holds the resident inputs and runs one pass of the path."""

    STAGES = ["sample_tuples", "shot_frames", "shot352", "shot_encoder", "encode_tuples", "tuple_mlp",
              "decode_bins", "vote_frames", "vote_center", "backvote_filter", "rot_bins", "scale_head", "assemble_pose", "gather"]

    def __init__(self, args, rank, world, dev):
        from cppf2_amd import dist as cdist
        from cppf2_amd import ops, synth
        from cppf2_amd.models import BeyondCPPFShot
        from cppf2_amd.pipeline import VotingPipeline
        self.ops, self.dist, self.args, self.rank, self.world, self.dev = ops, cdist, args, rank, world, dev
        assert cdist.shard(args.scenes_per_gpu * world, rank, world) == (rank * args.scenes_per_gpu, (rank + 1) * args.scenes_per_gpu)
        B, N, T = args.scenes_per_gpu, args.points, args.tuples
        self.B, self.N, self.T = B, N, T
        self.scene0 = rank * B
        scenes = [synth.make_scene(args.seed, self.scene0 + b, N) for b in range(B)]
        self.scenes = scenes
        self.pts = torch.from_numpy(np.concatenate([s["pc"] for s in scenes])).to(dev)
        self.pipe = VotingPipeline([N] * B, [T] * B, k=5, res=Cfg.res, num_rots=args.rots, cfg_up=Cfg.up,
                                   cfg_right=Cfg.right, cfg_front=Cfg.front, cells_cap=1 << 21,
                                   vote_mode=args.vote_mode, device=dev)
        torch.manual_seed(args.seed)
        self.model = BeyondCPPFShot(Cfg()).to(dev).eval()
        # teacher prior (untimed setup): peaked at the true canonical coordinates of each tuple's pair
        idx = ops.sample_tuples(N, T, 5, args.seed, tuple(range(self.scene0, self.scene0 + B)), dev)
        canon = torch.from_numpy(np.concatenate([s["pc_canon"] for s in scenes])).to(dev)
        base = (torch.arange(B, device=dev, dtype=torch.int64) * N).repeat_interleave(T)
        coords = canon[(idx[:, :2].long() + base[:, None]).reshape(-1)].reshape(B * T, 6)
        pos = (coords.clamp(-0.5, 0.5) + 0.5) * 31.0
        kbin = torch.arange(32, device=dev, dtype=torch.float32)
        self.prior = (-0.5 * ((kbin[None, None, :] - pos[..., None]) / 0.6) ** 2).contiguous()
        self.shot = torch.empty((B * N, 352), dtype=torch.float32, device=dev)
        self.normal = torch.empty((B * N, 3), dtype=torch.float32, device=dev)
        self.all_records = None
        self.records_buf = torch.empty((B * world, 160), dtype=torch.uint8, device=dev)      # the gather's result, allocated once
        self.scales_buf = torch.zeros((B * T, 3), dtype=torch.float32, device=dev)
        self.eager = bool(args.eager_scale_head)
        self.materialize = bool(getattr(args, "materialize_tuples", False))
        self.host_times = None          # debugging aid: host-side time stamps of the stage boundaries (CPPF_BENCH_HOSTTIMES=1)
        self.ev = None

    @property
    def gather(self):
        from cppf2_amd import models
        return (not self.materialize) and models.MLP_ARITH in ("split", "split16") and self.model.gather_supported(64, 5)

    EVENT_SLOTS = 8      # timed steps sampled for the per-stage HIP-event times (events created and first recorded before timing)

    def prepare_events(self):
        """One set of HIP events per sampled step, created and recorded once OUTSIDE the timed region: creating ~15 timing
        events per step inside it (round 1-2a) intermittently stalls the host for 30-40 ms a few dozen launches after a device
        synchronisation on this ROCm -- the chip idles, +1 ms per step averaged over a 30-step loop."""
        self.ev_pool = []
        for _ in range(self.EVENT_SLOTS):
            evs = [(n, torch.cuda.Event(enable_timing=True)) for n in ["start"] + self.STAGES]
            for _, e in evs:
                e.record()
            self.ev_pool.append(evs)

    def _mark(self, name):
        if self.host_times is not None:
            self.host_times.append((name, time.perf_counter()))
        if self.ev is not None:
            n, e = self.ev_pool[self.ev_slot][len(self.ev)]
            assert n == name
            e.record()
            self.ev.append((n, e))

    @torch.no_grad()
    def run(self, timed=None):
        """timed = None, or the slot (< EVENT_SLOTS) of the prepared event set this step records its stage boundaries in."""
        from cppf2_amd import shot as shotmod
        ops, pipe, a = self.ops, self.pipe, self.args
        B, N, T = self.B, self.N, self.T
        self.ev = [] if timed is not None else None
        self.ev_slot = timed
        self._mark("start")
        idx = ops.sample_tuples(N, T, 5, a.seed, tuple(range(self.scene0, self.scene0 + B)), self.dev)
        self._mark("sample_tuples")
        shotmod.prepare_device(self.pts, pipe.pt_off, Cfg.res * 10, Cfg.res * 10, self.normal)   # eval.py:210
        self._mark("shot_frames")
        shot = shotmod.describe_device(self.pts, pipe.pt_off, self.normal, Cfg.res * 10, out=self.shot,
                                       nan_to_zero=True)                                   # eval.py:215 folded in
        self._mark("shot352")
        normal = ops.nan_to_zero_(self.normal)                             # eval.py:216
        feat = self.model.encode_points(shot)
        self._mark("shot_encoder")
        eager = self.eager
        from cppf2_amd.models import decode_supported, fused_stack
        u = ops.philox_uniform(T, 6, a.seed, 1, tuple(range(self.scene0, self.scene0 + B)), self.dev)
        drawn = False
        if self.gather:
            # train_shot.py:75-83 without its 1.8 GB of rows: pair features + global indices, the first ResLayer gathers
            heads, gidx = ops.encode_tuples_shot_heads(self.pts, idx, normal, pipe.pt_off, pipe.tup_off)
            self._mark("encode_tuples")
            # tuple encoder + logit head: [gathered 360 -> 128 + 4 x 128] [128 -> 256 (tapped: the tuple features) + 2 x 256]
            # [256 -> 192 + bin draw]; eval.py:225-229 is the epilogue of the last kernel, the logits are never written
            drawn = decode_supported(self.model.logit_encoder, feat)
            res, tf = fused_stack((self.model.tuple_encoder, self.model.logit_encoder), None, gather=(heads, gidx, feat),
                                  decode=(u, self.prior, pipe.bins) if drawn else None)
            if not drawn:
                logits = res.reshape(tf.shape[0], 6, -1)
            feat = fused_stack(self.model.scale_encoder, tf) if eager else tf       # eager: the scale head on every tuple
        else:
            x = ops.encode_tuples_shot(self.pts, idx, feat, normal, pipe.pt_off, pipe.tup_off)
            self._mark("encode_tuples")
            logits, feat = self.model.heads(x, lazy_scale=not eager)
        self._mark("tuple_mlp")
        if drawn:
            pipe.decode_from_bins(self.pts, idx)
        else:
            pipe.decode(self.pts, idx, logits, u, prior=self.prior)      # teacher prior added inside the decode kernel
        self._mark("decode_bins")
        pipe.vote_center(self.pts, idx, phase=1)      # scene bounds + per-pair circle frames
        self._mark("vote_frames")
        pipe.vote_center(self.pts, idx, phase=2)      # the vote kernel (+ the 5 us final argmax)
        self._mark("vote_center")
        pipe.backvote(self.pts, idx)
        self._mark("backvote_filter")
        pipe.rot_bins(self.pts, idx)
        self._mark("rot_bins")
        if eager:
            scales = feat                                # heads() already ran the scale head on every tuple
        else:
            # the scale head is read only for the kept pairs (eval.py:272): run it on those rows (~10 % of the tuples).
            # (Round 2a ran it on a second stream beside the rotation votes; with the head as two short matrix-core kernels
            # the two orders take the same time -- 0.49 ms for both stages -- so it is in stream order: one stream, no waits.)
            # Every kernel of it is the library's: kept-row list, two gathered / split matrix-core layers, the 64 -> 3 layer with
            # the scatter into the [T, 3] buffer assemble() reads folded into its store.
            scales = self.model.scale_head_rows(feat, pipe.kept_rows32(), scatter=(pipe.kept_count, pipe.max_kept, self.scales_buf))
        self._mark("scale_head")
        pipe.assemble(scales)
        self._mark("assemble_pose")
        # the one collective of the path: 160-byte records of every rank's scenes, global scene order
        self.all_records = self.dist.gather_results(pipe.results, B * self.world, out=self.records_buf)
        self._mark("gather")
        return self.ev


class EnsembleStep(Step):
    """BASELINE configs[2]: one pass of eval.py:207-372 over the batch -- shared tuple table and SHOT descriptors, then the DINO
    model's pass and the SHOT model's pass (tuple MLP -> bin draw -> centre vote -> back-vote filter -> rotation votes -> scale
    head on the kept pairs -> pose -> alignment loss each), then the selection; every launch is a kernel of libcppf_hip.so.
    The DINOv2 descriptors are inputs of the path (seeded unit vectors [N, 1024], resident like the points)."""

    PASS = ["encode", "tuple_mlp", "decode_bins", "vote_frames", "vote_center", "backvote_filter", "rot_bins", "scale_head",
            "assemble_pose", "alignment_loss"]
    STAGES = (["sample_tuples", "shot_frames", "shot352", "shot_encoder", "dino_point_transforms"]
              + ["dino_" + n for n in PASS] + ["shot_" + n for n in PASS] + ["select", "gather"])

    def __init__(self, args, rank, world, dev):
        super().__init__(args, rank, world, dev)
        from cppf2_amd.models import BeyondCPPFDino
        torch.manual_seed(args.seed + 1)
        self.dino = BeyondCPPFDino(Cfg()).to(dev).eval()
        g = torch.Generator(device="cpu").manual_seed(args.seed + 17 + rank)
        self.desc = torch.nn.functional.normalize(torch.randn((self.B * self.N, 1024), generator=g), dim=-1).to(dev)
        self.scales_buf2 = torch.zeros((self.B * self.T, 3), dtype=torch.float32, device=dev)

    def _vote_pass(self, pre, model, tf, idx, scales_buf, pipe=None, before_loss=None):
        pipe = pipe or self.pipe
        pipe.decode_from_bins(self.pts, idx)
        self._mark(pre + "decode_bins")
        pipe.vote_center(self.pts, idx, phase=1)
        self._mark(pre + "vote_frames")
        pipe.vote_center(self.pts, idx, phase=2)
        self._mark(pre + "vote_center")
        pipe.backvote(self.pts, idx)
        self._mark(pre + "backvote_filter")
        pipe.rot_bins(self.pts, idx)
        self._mark(pre + "rot_bins")
        scales = model.scale_head_rows(tf, pipe.kept_rows32(), scatter=(pipe.kept_count, pipe.max_kept, scales_buf))
        self._mark(pre + "scale_head")
        pipe.assemble(scales)
        self._mark(pre + "assemble_pose")
        if before_loss is not None:
            before_loss()
        pipe.alignment_loss(self.pts, idx, True)          # bottle: up-symmetric, y only (eval.py:360-361)
        self._mark(pre + "alignment_loss")

    @torch.no_grad()
    def run_two_streams(self, streams):
        """The same step as run() the way eval.run_ensemble runs it: the DINO pass on streams[0], the SHOT descriptors + SHOT
        pass on streams[1] (twin pipeline: own working buffers, shared record slots), one event for the DINO pass' scale (it
        scores the SHOT pass too, eval.py:308-310), selection and gather on the calling stream.  No per-stage events."""
        from cppf2_amd import shot as shotmod
        from cppf2_amd.models import fused_stack
        ops, pipe, a = self.ops, self.pipe, self.args
        B, N, T = self.B, self.N, self.T
        ids = tuple(range(self.scene0, self.scene0 + B))
        if getattr(self, "pipe_b", None) is None:
            self.pipe_b = pipe.twin()
            self.dino_done = torch.cuda.Event()
        pipe_b = self.pipe_b
        self.ev = None
        main = torch.cuda.current_stream()
        idx = ops.sample_tuples(N, T, 5, a.seed, ids, self.dev)
        for st_ in streams:
            st_.wait_stream(main)
        with torch.cuda.stream(streams[0]):
            pipe.use_slot(0)
            fold = self.dino.first_layer_fold(5)
            tables = fold.tables(self.dino.transform_points(self.desc))
            u = ops.philox_uniform(T, 6, a.seed, 1, ids, self.dev)
            heads, gidx = ops.encode_tuples_coord_heads(self.pts, idx, pipe.pt_off, pipe.tup_off)
            _, tf = fused_stack((self.dino.tuple_encoder, self.dino.logit_encoder), None, gather=(heads, gidx, tables, fold),
                                decode=(u, self.prior, pipe.bins))
            self._vote_pass("dino_", self.dino, tf, idx, self.scales_buf, pipe=pipe, before_loss=self.dino_done.record)
        with torch.cuda.stream(streams[1]):
            shotmod.prepare_device(self.pts, pipe.pt_off, Cfg.res * 10, Cfg.res * 10, self.normal)
            shot = shotmod.describe_device(self.pts, pipe.pt_off, self.normal, Cfg.res * 10, out=self.shot, nan_to_zero=True)
            normal = ops.nan_to_zero_(self.normal)
            feat = self.model.encode_points(shot)
            pipe_b.use_slot(1)
            u2 = ops.philox_uniform(T, 6, a.seed, 2, ids, self.dev)
            heads2, gidx2 = ops.encode_tuples_shot_heads(self.pts, idx, normal, pipe.pt_off, pipe.tup_off)
            _, tf2 = fused_stack((self.model.tuple_encoder, self.model.logit_encoder), None, gather=(heads2, gidx2, feat),
                                 decode=(u2, self.prior, pipe_b.bins))
            self._vote_pass("shot_", self.model, tf2, idx, self.scales_buf2, pipe=pipe_b,
                            before_loss=lambda: torch.cuda.current_stream().wait_event(self.dino_done))
        for st_ in streams:
            main.wait_stream(st_)
        pipe.select(True, True)
        self.all_records = self.dist.gather_results(pipe.selected, B * self.world, out=self.records_buf)
        # the per-pass tensors were allocated on the side streams and are released here, on the calling stream: keep them alive
        # until the side streams are done with them (the next call's wait_stream orders the reuse)
        self._keep = (idx, tables, u, heads, gidx, tf, shot, feat, u2, heads2, gidx2, tf2)
        return None

    @torch.no_grad()
    def run(self, timed=None):
        from cppf2_amd import shot as shotmod
        from cppf2_amd.models import fused_stack
        ops, pipe, a = self.ops, self.pipe, self.args
        B, N, T = self.B, self.N, self.T
        ids = tuple(range(self.scene0, self.scene0 + B))
        self.ev = [] if timed is not None else None
        self.ev_slot = timed
        self._mark("start")
        idx = ops.sample_tuples(N, T, 5, a.seed, ids, self.dev)                          # eval.py:207: one table for both models
        self._mark("sample_tuples")
        shotmod.prepare_device(self.pts, pipe.pt_off, Cfg.res * 10, Cfg.res * 10, self.normal)   # eval.py:210
        self._mark("shot_frames")
        shot = shotmod.describe_device(self.pts, pipe.pt_off, self.normal, Cfg.res * 10, out=self.shot, nan_to_zero=True)
        self._mark("shot352")
        normal = ops.nan_to_zero_(self.normal)
        feat = self.model.encode_points(shot)
        self._mark("shot_encoder")
        # ---- model 0: DINO (train_dino.py:91-97, 128-133; eval.py:221) -------------------------------------------
        pipe.use_slot(0)
        fold = self.dino.first_layer_fold(5)
        tables = fold.tables(self.dino.transform_points(self.desc))      # desc_transform, then the folded slot products: per POINT
        self._mark("dino_point_transforms")
        u = ops.philox_uniform(T, 6, a.seed, 1, ids, self.dev)
        heads, gidx = ops.encode_tuples_coord_heads(self.pts, idx, pipe.pt_off, pipe.tup_off)
        self._mark("dino_encode")
        _, tf = fused_stack((self.dino.tuple_encoder, self.dino.logit_encoder), None, gather=(heads, gidx, tables, fold),
                            decode=(u, self.prior, pipe.bins))
        self._mark("dino_tuple_mlp")
        self._vote_pass("dino_", self.dino, tf, idx, self.scales_buf)
        # ---- model 1: SHOT (train_shot.py:75-83, 117-122; eval.py:223) -------------------------------------------
        pipe.use_slot(1)
        u = ops.philox_uniform(T, 6, a.seed, 2, ids, self.dev)
        heads, gidx = ops.encode_tuples_shot_heads(self.pts, idx, normal, pipe.pt_off, pipe.tup_off)
        self._mark("shot_encode")
        _, tf = fused_stack((self.model.tuple_encoder, self.model.logit_encoder), None, gather=(heads, gidx, feat),
                            decode=(u, self.prior, pipe.bins))
        self._mark("shot_tuple_mlp")
        self._vote_pass("shot_", self.model, tf, idx, self.scales_buf2)
        pipe.select(True, True)                                                           # eval.py:365-372
        self._mark("select")
        self.all_records = self.dist.gather_results(pipe.selected, B * self.world, out=self.records_buf)
        self._mark("gather")
        return self.ev


class DenseStep(Step):
    """BASELINE configs[4] (YCB-V instance-level: 65 536 pairs per scene, float16 per-point features, uncertainty-weighted centre
    votes -- extensions, pinned by the oracle's restatement only: tests/test_gpu_parity.py).  The SHOT model's path with the tuple
    rows materialised from the float16 table (cppf_encode_tuples_shot_f16) and per-pair vote weights in [0, 4] (fixed-point
    accumulator, cppf_vote_center's vote_wt); the weights here are a resident synthetic confidence per pair."""

    STAGES = ["sample_tuples", "shot_frames", "shot352", "shot_encoder", "cast_f16", "encode_tuples_f16", "tuple_mlp", "decode_bins",
              "vote_frames", "vote_center_weighted", "backvote_filter", "rot_bins", "scale_head", "assemble_pose", "gather"]

    def __init__(self, args, rank, world, dev):
        args.tuples = 65536
        super().__init__(args, rank, world, dev)
        ids = tuple(range(self.scene0, self.scene0 + self.B))
        self.vote_wt = (self.ops.philox_uniform(self.T, 1, args.seed, 7, ids, dev).reshape(-1) * 2.0).contiguous()    # untimed setup
        self.feat16 = torch.empty((self.B * self.N, 64), dtype=torch.float16, device=dev)

    @torch.no_grad()
    def run(self, timed=None):
        from cppf2_amd import shot as shotmod
        from cppf2_amd.models import fused_stack
        ops, pipe, a = self.ops, self.pipe, self.args
        B, N, T = self.B, self.N, self.T
        ids = tuple(range(self.scene0, self.scene0 + B))
        self.ev = [] if timed is not None else None
        self.ev_slot = timed
        self._mark("start")
        idx = ops.sample_tuples(N, T, 5, a.seed, ids, self.dev)
        self._mark("sample_tuples")
        shotmod.prepare_device(self.pts, pipe.pt_off, Cfg.res * 10, Cfg.res * 10, self.normal)
        self._mark("shot_frames")
        shot = shotmod.describe_device(self.pts, pipe.pt_off, self.normal, Cfg.res * 10, out=self.shot, nan_to_zero=True)
        self._mark("shot352")
        normal = ops.nan_to_zero_(self.normal)
        feat = self.model.encode_points(shot)
        self._mark("shot_encoder")
        ops.cast_f16(feat, out=self.feat16)
        self._mark("cast_f16")
        u = ops.philox_uniform(T, 6, a.seed, 1, ids, self.dev)
        x = ops.encode_tuples_shot(self.pts, idx, self.feat16, normal, pipe.pt_off, pipe.tup_off)      # [T, 360] float32 rows
        self._mark("encode_tuples_f16")
        _, tf = fused_stack((self.model.tuple_encoder, self.model.logit_encoder), x, decode=(u, self.prior, pipe.bins))
        self._mark("tuple_mlp")
        pipe.decode_from_bins(self.pts, idx)
        self._mark("decode_bins")
        pipe.vote_center(self.pts, idx, vote_wt=self.vote_wt, phase=1)
        self._mark("vote_frames")
        pipe.vote_center(self.pts, idx, vote_wt=self.vote_wt, phase=2)
        self._mark("vote_center_weighted")
        pipe.backvote(self.pts, idx)
        self._mark("backvote_filter")
        pipe.rot_bins(self.pts, idx)
        self._mark("rot_bins")
        scales = self.model.scale_head_rows(tf, pipe.kept_rows32(), scatter=(pipe.kept_count, pipe.max_kept, self.scales_buf))
        self._mark("scale_head")
        pipe.assemble(scales)
        self._mark("assemble_pose")
        self.all_records = self.dist.gather_results(pipe.results, B * self.world, out=self.records_buf)
        self._mark("gather")
        return self.ev


def report_dense(args, step, dt, evs, world, backend):
    """The JSON line of --workload dense64k (rank 0): contract fields, per-stage times, the two extension kernels' rooflines."""
    from cppf2_amd import models as _models
    B, N, T, R = step.B, step.N, step.T, args.rots
    stage_ms = {}
    for ev in evs:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            stage_ms[n1] = stage_ms.get(n1, 0.0) + e0.elapsed_time(e1) / len(evs)
    rec = step.pipe.results_to_numpy()
    all_rec = step.pipe.results_to_numpy(step.all_records)
    assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), rec.tobytes())
    ok = 0
    for b in range(B):
        sc = step.scenes[b]
        terr = np.linalg.norm(rec["t"][b] - sc["t"])
        cosang = abs(float(rec["R"][b][:, 1] @ sc["R"][:, 1]))
        ok += int(terr < 0.05 and np.degrees(np.arccos(min(cosang, 1.0))) < 5.0)
    nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
    ex, al = tuple_mlp_flops("shot", B, T, N, nprod)
    enc_bytes = B * (T * 5 * 4 + N * 24 + N * 64 * 2 + T * 360 * 4)                 # indices + points + normals + f16 table in, rows out
    enc_ms, vc_ms, mlp_ms = stage_ms["encode_tuples_f16"], stage_ms["vote_center_weighted"], stage_ms["tuple_mlp"]
    enc_c = counter_entry("encode_shot_f16_kernel")
    vc_act = unit_activity(counter_entry("vote_center_persist_kernel<true>"))
    total = B * world * args.steps
    line = {
        "metric": "scenes/sec (1/2/4/8 GPU) at 4096 pts x 20k tuples; 5deg5cm match vs ref",
        "value": total / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: dense pairs -- %d scenes/GPU x %d pts x %d tuples x %d rots, SHOT model, float16 "
                               "per-point feature table (tuple rows materialised from it), per-pair vote weights in [0, 4] on the "
                               "fixed-point centre accumulator (extensions the reference does not have); random-init weights + teacher "
                               "prior; MLP arithmetic %s; one HIP stream" % (B, N, T, R, _models.MLP_ARITH),
                   "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "parallelism": "scene-sharded x%d" % world},
        "pairs_per_s": total * T / dt,
        "roofline": dict(bound="mfma", kernel="tuple_mlp", kernel_name="reslayer_split_kernel", achieved=ex / 1e12 / (mlp_ms / 1e3),
                         peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=ex / 1e12 / (mlp_ms / 1e3) / BF16_MFMA_PEAK_TFLOPS,
                         traffic=None, launch_ms=mlp_ms, launches=3, frac_kind="executed_bf16_mfma" if nprod == 6.0 else "executed_fp16_mfma",
                         algorithmic_f32_flops_per_step=al,
                         per_kernel={
                             "encode_tuples_f16": dict(kernel="encode_shot_f16_kernel", bound="hbm", ms=round(enc_ms, 4), alg_MB=round(enc_bytes / 1e6, 1),
                                                       alg_GBs=round(enc_bytes / 1e9 / (enc_ms / 1e3), 1),
                                                       frac=round(enc_bytes / 1e9 / (enc_ms / 1e3) / HBM_PEAK_GBS, 4),
                                                       pmc_MB=None if hbm_bytes(enc_c) is None else round(hbm_bytes(enc_c) / 1e6, 1)),
                             "vote_center_weighted": dict(kernel="vote_center_persist_kernel<true>", bound="valu", ms=round(vc_ms, 4),
                                                          frac=None if vc_act is None else vc_act["valu_busy"], activity=vc_act,
                                                          work={"votes_per_launch": B * T * R, "votes_per_s": B * T * R / (vc_ms / 1e3)}),
                         },
                         counters=("this run's rocprofv3 passes" if "reason" not in COUNTERS else "null: " + str(COUNTERS.get("reason"))),
                         per_stage_ms={s_: round(stage_ms.get(s_, 0.0), 4) for s_ in DenseStep.STAGES}),
        "cpu_baseline": None, "cpu_baseline_note": "the extensions have no reference path to time; their oracle restatements are "
                                                   "checked in tests/test_gpu_parity.py",
        "pose_5deg5cm_vs_gt": ok / B,
        "collective": {"backend": backend or "none (one rank: the local records are the result)", "world": world,
                       "records_gathered": int(all_rec.shape[0])},
        "ok": True, "problems": [],
    }
    print(json.dumps(line))


def tuple_mlp_flops(model, B, T, N, nprod):
    """(executed MFMA flops, algorithmic float32 flops) of one pass' tuple MLP launches (the three reslayer_split launches; for
    the DINO model also the two per-point Linear launches), K padded to 16 in the executed figure."""
    def layer(k, n, proj):
        return 2.0 * n * ((k + 15) // 16 * 16) * (2 if proj else 1) + 2.0 * n * n, 2.0 * n * k * (2 if proj else 1) + 2.0 * n * n
    tail = [(128, 128, False)] * 4 + [(128, 256, True), (256, 256, False), (256, 256, False), (256, 192, True)]
    if model == "shot":
        layers = [(360, 128, True)] + tail
        ex = sum(layer(*l)[0] for l in layers) * B * T
        al = sum(layer(*l)[1] for l in layers) * B * T
        return nprod * ex, al
    # DINO: the first layer's products run over the 30 (32) coordinate columns per tuple; its descriptor columns (and
    # desc_pair_transform) are the per-point slot tables, desc_transform the per-point Linear in front of them
    layers = [(30, 128, True)] + tail
    ex = sum(layer(*l)[0] for l in layers) * B * T + (2.0 * 1024 * 256 + 2.0 * 256 * 1280) * B * N
    # algorithmic = the reference's network as written: 286-column rows, desc_transform on k gathered descriptors per tuple,
    # desc_pair_transform over their concatenation (train_dino.py:95-96)
    al = (sum(layer(*l)[1] for l in [(286, 128, True)] + tail) + 5 * 2.0 * 1024 * 256 + 2.0 * 1280 * 256) * B * T
    return nprod * ex, al


def ensemble_mlp_traffic():
    """HBM bytes per ensemble step of all matrix-core launches (every reslayer_split_kernel instantiation: both tuple MLPs, the
    point encoder, the DINO model's per-point Linear launches, the scale heads), from this run's counter passes."""
    passes = (COUNTERS.get("ensemble_select_kernel") or {}).get("launches")
    if not passes:
        return None
    tot = 0.0
    for k_, v in COUNTERS.items():
        if isinstance(v, dict) and k_.startswith("reslayer_split_kernel"):
            b_ = hbm_bytes(v)
            if b_ is None:
                return None
            tot += b_ * v["launches"]
    return tot / passes


def report_ensemble(args, step, dt, evs, world, backend):
    """The JSON line of --workload ensemble (rank 0): the contract fields + roofline + cpu_baseline, per model."""
    from cppf2_amd import models as _models
    B, N, T, R = step.B, step.N, step.T, args.rots
    stage_ms = {}
    for ev in evs:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            stage_ms[n1] = stage_ms.get(n1, 0.0) + e0.elapsed_time(e1) / len(evs)
    step_times = sorted(ev[0][1].elapsed_time(ev[-1][1]) for ev in evs)
    shared = ["sample_tuples", "shot_frames", "shot352"]
    dino_ms = stage_ms.get("dino_point_transforms", 0.0) + sum(stage_ms.get("dino_" + n, 0.0) for n in EnsembleStep.PASS)
    shot_ms = stage_ms.get("shot_encoder", 0.0) + sum(stage_ms.get("shot_" + n, 0.0) for n in EnsembleStep.PASS)
    shared_ms = sum(stage_ms.get(n, 0.0) for n in shared)
    nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
    ex_d, al_d = tuple_mlp_flops("dino", B, T, N, nprod)
    ex_s, al_s = tuple_mlp_flops("shot", B, T, N, nprod)
    mlp_ms = stage_ms["dino_point_transforms"] + stage_ms["dino_tuple_mlp"] + stage_ms["shot_tuple_mlp"]
    rec = step.pipe.results_to_numpy(step.pipe.selected)
    all_rec = step.pipe.results_to_numpy(step.all_records)
    assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), rec.tobytes())
    both = [step.pipe.results_to_numpy(step.pipe.result_slots[m]) for m in (0, 1)]
    ok = 0
    for b in range(B):
        sc = step.scenes[b]
        terr = np.linalg.norm(rec["t"][b] - sc["t"])
        cosang = abs(float(rec["R"][b][:, 1] @ sc["R"][:, 1]))
        ok += int(terr < 0.05 and np.degrees(np.arccos(min(cosang, 1.0))) < 5.0)
    pick = rec["pad_"][:, 0]
    cpu = None
    agree = None
    if args.cpu_scenes > 0 and world == 1:
        from oracle import pipeline_oracle as PO         # the checker, timed as the CPU baseline (never the product path)
        from oracle import cppf_oracle as O
        from oracle import shot_oracle as S
        from cppf2_amd import synth
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
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count()
        cpu = dict(value=n_cpu / dtc, unit="scenes/s", cores=cores, kind="port",
                   sample="%d instance(s) of the same workload, both models (NumPy oracle: mlp_dino + mlp_shot + run_instance_ensemble; "
                          "C SHOT oracle), %.1f s" % (n_cpu, dtc),
                   threads_per_stage={"shot_descriptor": 1, "mlp_matmuls": "BLAS default (all cores)", "votes_and_bins": 1})
        agree = dict(instances=n_cpu,
                     pick_equal=int(sum(int(pick[b]) == o["pick"] for b, o in enumerate(outs))),
                     centre_argmax_equal=[int(sum(int(both[m]["argmax"][b]) == o["models"][m]["argmax"] for b, o in enumerate(outs))) for m in (0, 1)],
                     up_bin_equal=[int(sum(int(both[m]["up_idx"][b]) == o["models"][m]["up_idx"] for b, o in enumerate(outs))) for m in (0, 1)],
                     max_abs_loss_difference=float(max(abs(float(step.pipe.losses[m][b]) - o["models"][m]["loss"])
                                                       for b, o in enumerate(outs) for m in (0, 1))))
    total = B * world * args.steps
    line = {
        "metric": "scenes/sec (1/2/4/8 GPU) at 4096 pts x 20k tuples; 5deg5cm match vs ref",
        "value": total / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: DINO + SHOT ensemble, every instance voted by BOTH models (eval.py:219-372), %d "
                               "instances/GPU x %d pts x %d tuples x %d rots, 720 sphere bins, res 2 mm, bottle axes (y-only "
                               "alignment loss); random-init weights + teacher prior; DINOv2 descriptors = seeded unit vectors "
                               "[N, 1024] resident in HBM (inputs of the path); MLP arithmetic %s; one scene = one instance "
                               "through both models" % (B, N, T, R, _models.MLP_ARITH),
                   "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "parallelism": "scene-sharded x%d" % world},
        "step_ms_sampled": {"min": round(step_times[0], 4), "median": round(step_times[len(step_times) // 2], 4),
                            "max": round(step_times[-1], 4), "n": len(step_times)},
        "per_model_ms": {"shared (sampler, normals + SHOT352)": round(shared_ms, 4),
                         "dino (point transforms .. alignment loss)": round(dino_ms, 4),
                         "shot (point encoder .. alignment loss)": round(shot_ms, 4),
                         "dino_over_shot": round(dino_ms / shot_ms, 4) if shot_ms > 0 else None},
        "picked": {"dino": int((pick == 0).sum()), "shot": int((pick == 1).sum()), "none": int((pick < 0).sum())},
        "roofline": dict(bound="mfma", kernel="tuple_mlp (both models) + the DINO model's per-point Linear launches",
                         kernel_name="reslayer_split_kernel", achieved=(ex_d + ex_s) / 1e12 / (mlp_ms / 1e3),
                         peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=(ex_d + ex_s) / 1e12 / (mlp_ms / 1e3) / BF16_MFMA_PEAK_TFLOPS,
                         traffic=ensemble_mlp_traffic(),
                         traffic_source=("rocprofv3 counter passes of this run: HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) of every "
                                         "reslayer_split_kernel launch of a step" if "reason" not in COUNTERS
                                         else "null: " + str(COUNTERS.get("reason"))),
                         launch_ms=mlp_ms, launches=8,
                         frac_kind="executed_bf16_mfma" if nprod == 6.0 else "executed_fp16_mfma",
                         executed_flops_per_step={"dino": ex_d, "shot": ex_s},
                         algorithmic_f32_flops_per_step={"dino (the reference's row form)": al_d, "shot": al_s},
                         per_stage_ms={s_: round(stage_ms.get(s_, 0.0), 4) for s_ in EnsembleStep.STAGES}),
        "cpu_baseline": cpu, "oracle_agreement": agree, "pose_5deg5cm_vs_gt": ok / B,
        "collective": {"backend": backend or "none (one rank: the local records are the result)", "world": world,
                       "records_gathered": int(all_rec.shape[0]), "bytes_per_rank": int(B * 160),
                       "gather_us": round(1e3 * stage_ms.get("gather", 0.0), 2)},
        "two_streams": getattr(step, "two", None),
        "value_single_stream": (step.two or {}).get("value_single_stream") if getattr(step, "two", None) else None,
        "ok": not (getattr(step, "two", None) and not step.two["records_identical_to_single_stream"]),
        "problems": (["two_streams: records differ from the single-stream ones"]
                     if (getattr(step, "two", None) and not step.two["records_identical_to_single_stream"]) else []),
    }
    print(json.dumps(line))


GATHERED_TUPLES = False         # set by main(): the encode stage writes pair features + indices only
FUSED_DRAW = False              # set by main(): the bins are drawn inside the MLP's output layer, the decode stage starts from them


def algorithmic_bytes(stage, B, N, T, R, S, G):
    """Compulsory bytes one launch of the stage's kernel moves for B scenes (SURVEY.md 8d per-scene figures)."""
    if stage == "encode_tuples" and GATHERED_TUPLES:
        return (T * 5 * 4 + N * 12 + N * 12 + T * 40 * 4 + T * 5 * 4) * B
    if stage == "decode_bins" and FUSED_DRAW:
        return (T * 6 * 4 + T * 8 + N * 12 + T * (12 + 24 + 4 + 24)) * B
    per_scene = {
        "sample_tuples": T * 5 * 4,
        "shot_frames": N * 12 * 2 + N * 12 + N * 19 * 8 * 2 + N * 56,   # points in+sorted, normals out, sums w+r, frames
        "shot352": N * 12 + N * 12 + N * 56 + N * 352 * 4,
        "encode_tuples": T * 5 * 4 + N * 12 + N * 12 + N * 64 * 4 + T * 360 * 4,
        "decode_bins": 2 * T * 6 * 32 * 4 + T * 6 * 4 + T * 8 + T * (8 + 12 + 24 + 4 + 24),   # logits + prior read
        # idx + tr + points, grid clear G*4, one 4-byte accumulator update per vote (V = T*R), argmax
        # read G*4.  (This implementation keeps the accumulator in LDS slabs, so its HBM traffic -- `traffic` -- is
        # well below this figure: the frames workspace and the per-slab re-reads of it.)
        "vote_center": T * 8 + T * 8 + N * 12 + G * 4 + T * R * 4 + G * 4,
        "vote_frames": T * 8 + T * 8 + N * 12 + T * 48,
        "backvote_filter": T * 8 + T * 8 + N * 12 + T * (1 + 4 + 4 + 8 + 4),
        "rot_bins": 2 * (T // 10) * (4 + 8 + 4 + 8 + 12) + 2 * S * 4,
        "assemble_pose": 160,
    }
    return per_scene.get(stage, 0) * B


STAGE_KERNEL = {"sample_tuples": "sample_tuples_kernel", "shot_frames": "shot_cov_kernel", "shot352": "shot_hist_kernel",
                "encode_tuples": "encode_shot_kernel<5, 16>", "decode_bins": "decode_bins_kernel<32>",
                "vote_frames": "vote_frames_kernel", "vote_center": "vote_center_persist_kernel", "backvote_filter": "backvote_kernel",
                "rot_bins": "rot_bins_lut_kernel<2>", "assemble_pose": "assemble_pose_kernel"}


# ---------------------------------------------------------------------------------------------------------------------
# Hardware counters measured IN THIS RUN: before this process touches the GPU, rank 0 of a one-rank run starts fresh child
# processes of itself under rocprofv3 (the program itself after `--`; counters in their own passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE KB: the guide's
# gfx950 correction) and one pass of SQ counters (VALU / LDS / matrix-pipe activity per kernel).  No rocprofv3, a failed pass or
# --no-counters: the fields are null with the reason -- never a number read from profiles/.
# ---------------------------------------------------------------------------------------------------------------------
COUNTER_PASSES = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
                  "SQ": ["SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU",
                         "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]}
COUNTERS = {"reason": "not collected"}


def kernel_key(name):
    """'void reslayer_split_kernel<4, true, ...>(float const*, ...)' -> 'reslayer_split_kernel<4, true, ...>'"""
    n = name.replace("void ", "")
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i].strip()
    return n.strip()


def collect_counters(argv_workload, passes=("FETCH_SIZE", "WRITE_SIZE", "SQ")):
    """Runs `rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py <workload flags> --steps 2 --warmup 1 --counter-child` once per
    pass and returns {kernel key [#large | #small]: {counter: mean per launch, "launches": n, "avg_us": mean duration}} -- a kernel
    launched both for all tuples and for the kept pairs is split into two duration classes."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"reason": "rocprofv3 not found on PATH or under /opt/rocm/bin"}
    out, fail = {}, []
    tmp = tempfile.mkdtemp(prefix="cppf_bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in list(env):
        if k_ in _SET_HERE:
            env.pop(k_)
    child = [sys.executable, os.path.abspath(__file__)] + argv_workload + ["--steps", "2", "--warmup", "1", "--counter-child"]
    for pname in passes:
        d = os.path.join(tmp, pname)
        cmd = [exe, "--kernel-trace", "--pmc"] + COUNTER_PASSES[pname] + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
        except Exception as e:      # noqa: BLE001
            fail.append("%s: %r" % (pname, e))
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            fail.append("%s: rocprofv3 exit %d, %d counter files; %s" % (pname, r.returncode, len(files), r.stderr.decode("utf-8", "replace")[-300:]))
            continue
        rows = [row for f in files for row in csv.DictReader(open(f))]
        dur = collections.defaultdict(dict)
        for row in rows:
            dur[kernel_key(row["Kernel_Name"])][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        cls = {}
        for k_, dd in dur.items():
            mx = max(dd.values())
            if "reslayer_split_kernel" in k_ and min(dd.values()) < 0.25 * mx:
                for i_, t_ in dd.items():
                    cls[(k_, i_)] = k_ + ("#large" if t_ >= 0.25 * mx else "#small")
        agg, cnt, us = collections.defaultdict(float), collections.defaultdict(set), collections.defaultdict(dict)
        for row in rows:
            k0 = kernel_key(row["Kernel_Name"])
            k_ = cls.get((k0, row["Dispatch_Id"]), k0)
            agg[(k_, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[k_].add(row["Dispatch_Id"])
            us[k_][row["Dispatch_Id"]] = dur[k0][row["Dispatch_Id"]] / 1e3
        for (k_, c_), v_ in agg.items():
            e = out.setdefault(k_, {})
            e[c_] = v_ / len(cnt[k_])
            e["launches"] = len(cnt[k_])
            e.setdefault("avg_us", {})[pname] = sum(us[k_].values()) / len(us[k_])
    shutil.rmtree(tmp, ignore_errors=True)
    if fail:
        out["reason"] = "; ".join(fail)
    return out


def counter_entry(name):
    """The counters of the kernel whose key starts with `name` (exact key first)."""
    if not name:
        return None
    if name in COUNTERS:
        return COUNTERS[name]
    hits = [v for k_, v in sorted(COUNTERS.items()) if isinstance(v, dict) and k_.startswith(name)]
    return hits[0] if hits else None


def hbm_bytes(entry):
    """HBM bytes per launch from one kernel's counters: 2 x FETCH_SIZE + WRITE_SIZE (KB), or None."""
    if not entry or "FETCH_SIZE" not in entry or "WRITE_SIZE" not in entry:
        return None
    return (2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0


# what limits each stage's kernel: "hbm" = streaming (bytes / time against 8 TB/s), "unit" = an execution unit (VALU or LDS: which
# one, and how busy, comes from the SQ counters), "latency" = one workgroup per scene or a chain of dependent phases
STAGE_BOUND = {"sample_tuples": "hbm", "encode_tuples": "hbm", "decode_bins": "hbm", "vote_frames": "hbm",
               "shot_frames": "unit", "shot352": "unit", "vote_center": "unit", "rot_bins": "unit",
               "backvote_filter": "latency", "assemble_pose": "latency"}


def pmc_traffic(stage):
    return hbm_bytes(counter_entry(STAGE_KERNEL.get(stage, "")))


def unit_activity(entry):
    """Busy fractions of the VALU, LDS and matrix pipes over one launch (rocprof's derived-metric definitions: VALUBusy =
    4 SQ_ACTIVE_INST_VALU / SIMDs / shader cycles; LDS = SQ_LDS_IDX_ACTIVE / CUs / shader cycles; GRBM_GUI_ACTIVE is summed
    over the 8 XCDs), and the bound they name."""
    if not entry or "GRBM_GUI_ACTIVE" not in entry:
        return None
    cyc = entry["GRBM_GUI_ACTIVE"] / 8.0
    if cyc <= 0:
        return None
    valu = 4.0 * entry.get("SQ_ACTIVE_INST_VALU", 0.0) / 1024.0 / cyc
    mfma = entry.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
    lds = entry.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256.0 / cyc
    conf = entry.get("SQ_LDS_BANK_CONFLICT", 0.0) / 256.0 / cyc
    us = entry.get("avg_us", {}).get("SQ")
    # (rocprof's VALUBusy counts 4 cycles per active VALU instruction-quad: a kernel that keeps the VALU saturated with
    # instructions that issue faster can read a few percent above 1)
    return dict(valu_busy=round(valu, 4), lds_busy=round(lds, 4), lds_bank_conflict=round(conf, 4), mfma_busy=round(mfma, 4),
                shader_clock_ghz=round(cyc / us / 1e3, 3) if us else None, valu_insts_per_launch=entry.get("SQ_INSTS_VALU"))


TUPLE_MLP_KERNELS = ("reslayer_split_kernel<4, true, true, false, 3, 0>#large", "reslayer_split_kernel<8, true, false, false, 3, 0>",
                     "reslayer_split_kernel<6, true, false, true, 3, 0>")


def pmc_traffic_mlp(pieces=3):
    """HBM bytes per step of the tuple MLP's three cppf_reslayer_split launches (the gathered 360 -> 128 chain; 128 -> 256 with
    the two 256-wide identity layers behind it; 256 -> 192 + bin draw: the launches `launch_ms` times), from this run's counter
    passes; the gathering kernel also runs the scale head's first layer on the kept pairs, a ~20 x shorter launch kept under
    its own key.  None when the passes did not run."""
    tot = 0.0
    for k_ in TUPLE_MLP_KERNELS:
        k_ = k_.replace(", 3, 0>", ", %d, 0>" % pieces)
        b_ = hbm_bytes(COUNTERS.get(k_) or COUNTERS.get(k_.replace("#large", "")))
        if b_ is None:
            return None
        tot += b_
    return tot


def cpu_baseline(args, step):
    """Times the oracle (NumPy + C SHOT) on the host for a bounded sample of the same workload."""
    from oracle import pipeline_oracle as PO
    from cppf2_amd import synth
    if args.cpu_scenes <= 0:
        return None
    weights = {k: v.detach().cpu().numpy() for k, v in step.model.state_dict().items()}
    trig = (step.pipe.cs.cpu().numpy(), step.pipe.sn.cpu().numpy())
    t0 = time.perf_counter()
    agree = []
    for b in range(args.cpu_scenes):
        sc = step.scenes[b]
        out = PO.run_scene_full(weights, sc["pc"], args.seed, step.scene0 + b, args.tuples, res=Cfg.res,
                                num_rots=args.rots, trig=trig,
                                prior_fn=lambda idx, sc=sc: synth.teacher_logits(sc["pc_canon"], idx, 32, 0.6))
        agree.append(out)
    dt = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    return dict(value=args.cpu_scenes / dt, unit="scenes/s", cores=cores, kind="port",
                sample="%d scene(s) of the same workload (first scenes of rank 0's batch), NumPy oracle + C SHOT "
                       "oracle (SHOT single-threaded like PCL, matmuls on all cores), %.1f s" % (args.cpu_scenes, dt),
                threads_per_stage={"shot_descriptor (C oracle, like PCL)": 1, "mlp_matmuls (NumPy -> BLAS)": "all cores (BLAS default)",
                                   "decode / votes / back-vote / rotation bins (NumPy)": 1}), agree


@torch.no_grad()
def arithmetic_evidence(step, rows_err=4096, scenes_flip=10):
    """Untimed evidence for the line's `dtype` and parity claims, measured on the bench's own tuples (rank 0, after timing):
    * mlp_error_vs_f64: the tuple MLP's logits (tuple_encoder + logit_encoder, train_shot.py:56-66) of the first `rows_err`
      tuples in split arithmetic (what the step runs) and on PyTorch's float32 library GEMMs, each against a float64 evaluation
      of the same weights and rows; errors relative to the largest |logit|;
    * bin_flip_rate_vs_expf: the bins the decode kernel draws (softmax_exp: hardware exp2, ~1.5 ulp) against the same draw
      with torch.exp (libm-accurate expf) in the same float32 running-sum order, over `scenes_flip` scenes x T tuples x 6."""
    import copy
    from cppf2_amd import models as M
    ops, pipe, a, dev = step.ops, step.pipe, step.args, step.dev
    N, T = step.N, step.T
    sc = min(scenes_flip, step.B)
    ids = tuple(range(step.scene0, step.scene0 + sc))
    idx = ops.sample_tuples(N, T, 5, a.seed, ids, dev)
    pt_off, tup_off = ops._uniform_offsets(N, sc, dev), ops._uniform_offsets(T, sc, dev)
    feat = step.model.encode_points(step.shot[:sc * N])
    x = ops.encode_tuples_shot(step.pts[:sc * N], idx, feat, step.normal[:sc * N], pt_off, tup_off)
    out = {}
    if M.MLP_ARITH in ("split", "split16"):
        arith0 = M.MLP_ARITH
        xe = x[:rows_err].contiguous()
        m64 = copy.deepcopy(step.model).double()
        want = m64.logit_encoder(m64.tuple_encoder(xe.double()))
        scale = want.abs().max().item()
        M.MLP_ARITH = "split"
        split = M.fused_stack(step.model.logit_encoder, M.fused_stack(step.model.tuple_encoder, xe.clone()))
        M.MLP_ARITH = "split16"
        split16 = M.fused_stack(step.model.logit_encoder, M.fused_stack(step.model.tuple_encoder, xe.clone()))
        M.MLP_ARITH = arith0
        native = step.model.logit_encoder(step.model.tuple_encoder(xe))              # plain nn.Linear: library f32 GEMMs

        def err(t):
            d = t.double() - want
            return {"max": d.abs().max().item() / scale, "rms": d.pow(2).mean().sqrt().item() / scale}
        out["mlp_error_vs_f64"] = {"rows": int(xe.shape[0]), "logit_scale": scale, "split_bf16x3": err(split),
                                   "split_f16x2": err(split16), "library_f32_gemm": err(native),
                                   "note": "relative to max |logit|; the timed step runs " + ("split_bf16x3" if arith0 == "split" else "split_f16x2")}
    logits = step.model.heads(x, lazy_scale=True)[0].contiguous()                  # [sc*T, 6, 32]
    u = ops.philox_uniform(T, 6, a.seed, 1, ids, dev)
    prior = step.prior[:sc * T]
    got = ops.decode_bins(logits, u, step.pts[:sc * N], idx, Cfg.up, Cfg.front, Cfg.right, pt_off, tup_off, prior=prior)["bins"]
    e = logits + prior
    p = torch.exp(e - e.max(-1, keepdim=True).values)
    cdf = torch.empty_like(p)
    run = torch.zeros_like(p[..., 0])
    for j in range(p.shape[-1]):                                                   # the kernel's float32 running sum, in bin order
        run = run + p[..., j]
        cdf[..., j] = run
    target = u.reshape(-1, 6) * run
    ref = (cdf <= target[..., None]).sum(-1).clamp(max=p.shape[-1] - 1).to(torch.int32)
    flips = int((ref != got).sum().item())
    out["bin_flip_rate_vs_expf"] = {"draws": int(ref.numel()), "flips": flips, "rate": flips / ref.numel(),
                                    "max_bin_distance": int((ref - got).abs().max().item())}
    return out


def cdist_forced():
    return os.environ.get("CPPF_DIST_FORCE_COLLECTIVE", "0") not in ("", "0")


def pin_rank_to_cores(local_rank, local_world):
    """One contiguous slice of the process' allowed cores per local rank (rank r of W gets cores [r c / W, (r + 1) c / W) of the
    sorted list): the host threads of a rank -- launch loop, RCCL proxy, the allocator -- stay on one socket's cores instead of
    migrating across all of them while eight ranks launch ~30 kernels per 12 ms step each.  GPUs 0..3 / 4..7 hang off sockets
    0 / 1 on the 8-GPU boards, and core ids are socket-major, so contiguous slices in LOCAL_RANK order are NUMA-local as well.
    CPPF_BENCH_NO_AFFINITY=1 leaves the affinity alone.  Returns the slice (or None)."""
    if local_world <= 1 or os.environ.get("CPPF_BENCH_NO_AFFINITY"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except (AttributeError, OSError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started without a launcher environment: run N FRESH rank processes (one per GPU,
    the environment torch.distributed.run would give them) and exit with their status; rank 0 prints the JSON line on the
    inherited stdout.  This process never initialises the GPU (device_count() does not) and never re-execs itself."""
    import socket
    import subprocess
    n = args.gpus
    backend = os.environ.get("CPPF_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if have < n and backend == "nccl":
        print("bench.py: --gpus %d but %d GPU(s) are visible; one rank per GPU over RCCL needs %d (CPPF_BENCH_BACKEND=gloo "
              "is the dry-run switch that lets ranks share a GPU)" % (n, have, n), file=sys.stderr)
        return 2
    if have < 1:
        print("bench.py needs a GPU", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % have), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        for k_ in _SET_HERE:                     # every rank sets up its own TunableOp table (its own device ordinal and directory)
            env.pop(k_, None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd()))
    # wait for the ranks without polling: the parent sleeps in waitpid until a child exits
    rc = 0
    live = {p_.pid: p_ for p_ in procs}
    while live:
        try:
            pid, status = os.waitpid(-1, 0)
        except ChildProcessError:
            break
        p_ = live.pop(pid, None)
        if p_ is None:
            continue
        r_ = os.waitstatus_to_exitcode(status)
        p_.returncode = r_
        if r_ != 0 and rc == 0:
            rc = r_ if r_ > 0 else 1
            for q_ in live.values():       # a failed rank leaves the others waiting in a collective: stop exactly those PIDs
                q_.terminate()
    return rc


def main():
    args = parse()
    if args.workload == "dense64k" and "--scenes-per-gpu" not in sys.argv:
        args.scenes_per_gpu = 16
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
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
        args.cpu_scenes = 0
    profiled = any(k_.startswith(("ROCP_", "ROCPROF")) for k_ in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled and not args.no_counters:
        # this process is itself being profiled (the tool's library is loaded and has initialised the GPU): no children from here
        args.no_counters = True
        COUNTERS["reason"] = "running under a profiler: the counter passes are started by unprofiled runs only"
    if world == 1 and rank == 0 and not args.no_counters and not cdist_forced():
        # BEFORE this process initialises the GPU: fresh children under rocprofv3, one counter pass each
        wl = ["--scenes-per-gpu", str(args.scenes_per_gpu), "--points", str(args.points), "--tuples", str(args.tuples), "--rots",
              str(args.rots), "--seed", str(args.seed), "--vote-mode", str(args.vote_mode), "--workload", args.workload]
        wl += ["--mlp-arith", args.mlp_arith] if args.mlp_arith else []
        wl += ["--eager-scale-head"] if args.eager_scale_head else []
        wl += ["--materialize-tuples"] if args.materialize_tuples else []
        COUNTERS.clear()
        COUNTERS.update(collect_counters(wl))
    elif args.no_counters:
        COUNTERS.setdefault("reason", "--no-counters")
        if COUNTERS["reason"] == "not collected":
            COUNTERS["reason"] = "--no-counters"
    else:
        COUNTERS["reason"] = "counter passes run at one rank only (N = 1)"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    affinity = pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    from cppf2_amd import dist as cdist
    backend = None
    if world > 1 or cdist.force_collective():
        # RCCL over xGMI; CPPF_BENCH_BACKEND=gloo is a dry-run switch for boxes with fewer GPUs than ranks.
        # CPPF_DIST_FORCE_COLLECTIVE=1 makes a one-rank run create the group and issue the all_gather too.
        backend = os.environ.get("CPPF_BENCH_BACKEND", "nccl")
        cdist.init(backend=backend, device=dev if backend == "nccl" else None)
        assert torch.distributed.get_world_size() == world and torch.distributed.get_rank() == rank

    from cppf2_amd import models as _models
    if args.mlp_arith:
        _models.MLP_ARITH = args.mlp_arith
    if args.workload in ("ensemble", "dense64k"):
        assert _models.MLP_ARITH in ("split", "split16"), "--workload %s runs the library's kernels (split arithmetic)" % args.workload
        step = EnsembleStep(args, rank, world, dev) if args.workload == "ensemble" else DenseStep(args, rank, world, dev)
        step.prepare_events()
        step.run()
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step.run()
        n_s = min(args.steps, Step.EVENT_SLOTS)
        sampled = {int(round((j + 0.5) * args.steps / n_s - 0.5)): j for j in range(n_s)}

        def sync_():
            torch.cuda.synchronize()
            if torch.distributed.is_initialized():
                torch.distributed.barrier()
                torch.cuda.synchronize()

        def loop_(fn):
            sync_()
            t0_ = time.perf_counter()
            evs_ = []
            for i_ in range(args.steps):
                ev_ = fn(i_)
                if ev_ is not None:
                    evs_.append(ev_)
            sync_()
            tm_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
            if torch.distributed.is_initialized():
                if torch.distributed.get_backend() == "gloo":
                    tm_ = tm_.cpu()
                torch.distributed.all_reduce(tm_, op=torch.distributed.ReduceOp.MAX)
            return float(tm_.item()), evs_
        # single stream: per-stage events, kernel durations
        dt1, evs = loop_(lambda i_: step.run(timed=sampled.get(i_)))
        step.two = None
        dt = dt1
        if args.workload == "ensemble" and not args.single_stream:
            # the product's batch mode (eval.run_ensemble): the two model passes on two HIP streams with twin pipelines
            torch.cuda.synchronize()
            ref_sel = step.pipe.selected.clone()
            ref_slots = step.pipe.result_slots.clone()
            streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for _ in range(max(2, args.warmup)):
                step.run_two_streams(streams)
            dt, _ = loop_(lambda i_: step.run_two_streams(streams))
            same = bool(torch.equal(step.pipe.selected, ref_sel) and torch.equal(step.pipe.result_slots, ref_slots))
            step.two = {"streams": 2, "records_identical_to_single_stream": same, "value_single_stream": step.B * world * args.steps / dt1,
                        "ms_per_step_single_stream": 1e3 * dt1 / args.steps,
                        "note": "the DINO pass and the SHOT pass (descriptors included) on two HIP streams, twin pipelines, one event "
                                "for the DINO scale; per-stage times come from the single-stream loop of the same run"}
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if rank == 0:
            (report_ensemble if args.workload == "ensemble" else report_dense)(args, step, float(tmax.item()), evs, world, backend)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    step = Step(args, rank, world, dev)
    global GATHERED_TUPLES, FUSED_DRAW
    with torch.no_grad():           # the support checks look at the inference mode Step.run() executes in
        GATHERED_TUPLES = step.gather
        if GATHERED_TUPLES:
            STAGE_KERNEL["encode_tuples"] = "encode_shot_heads_tile_kernel"
            from cppf2_amd.models import decode_supported
            FUSED_DRAW = decode_supported(step.model.logit_encoder, torch.empty((1, 256), device=dev))
            if FUSED_DRAW:
                STAGE_KERNEL["decode_bins"] = "decode_targets_kernel"
    step.prepare_events()
    step.run()                      # part of the untimed setup: allocator pools of both streams, GEMM solution table,
    torch.cuda.synchronize()        # kernel attributes -- so that even --warmup 0 times steady-state steps
    for _ in range(args.warmup):
        step.run()

    def sync():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # one completion event per step of the headline loops (recorded on the step's stream right after its last launch): the
    # intervals between consecutive completions are the per-step figures of the line (step_interval_ms)
    end_pool = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    step_ends = {}

    def completion_intervals(ends, group=1):
        """ms per step between step completions.  group = number of streams: concurrent steps run side by side and complete
        within a millisecond of each other, so the figure is the time from completion i to completion i + group, divided by group."""
        if len(ends) < group + 2:
            return None
        t = sorted(ends[0].elapsed_time(e) for e in ends)
        d = sorted((b_ - a_) / group for a_, b_ in zip(t[:-group], t[group:]))
        return {"min": round(d[0], 4), "median": round(d[len(d) // 2], 4), "max": round(d[-1], 4), "n": len(d),
                "p05": round(d[int(0.05 * len(d))], 4), "p95": round(d[int(0.95 * len(d))], 4), "steps_per_interval": group}

    def timed_loop(k, sample=True, pair=None):
        """EXACTLY k steps between two barrier + synchronize pairs; max over ranks of the wall-clock seconds.
        pair = ([step_a, step_b], [stream_a, stream_b]): step i runs on stream i & 1 with that stream's own state."""
        # per-stage times come from up to EVENT_SLOTS steps spread evenly over the (headline) loop
        n_s = min(k, Step.EVENT_SLOTS) if sample else 0
        sampled = {int(round((j + 0.5) * k / n_s - 0.5)): j for j in range(n_s)}
        sync()
        t0 = time.perf_counter()
        evs_ = []
        ends_ = step_ends[id(pair)] = []
        for i_ in range(k):
            if pair is None:
                ev_ = step.run(timed=sampled.get(i_))
                if sample:
                    end_pool[i_].record()
            else:
                with torch.cuda.stream(pair[1][i_ % len(pair[1])]):
                    ev_ = pair[0][i_ % len(pair[1])].run(timed=sampled.get(i_))
                    if sample:
                        end_pool[i_].record()
            if sample:
                ends_.append(end_pool[i_])
            if ev_ is not None:
                evs_.append(ev_)
        sync()
        dt_ = time.perf_counter() - t0
        tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
        if torch.distributed.is_initialized():
            if torch.distributed.get_backend() == "gloo":
                tmax = tmax.cpu()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), evs_

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
    # ---- the headline loop.  Default (round 4): the product's batch mode -- consecutive steps, i.e. independent scene batches,
    # alternate between TWO HIP streams, each with its own resident state (software pipelining: one batch's descriptor, voting and
    # small MLP kernels run beside the other batch's wide matrix-core kernels).  Both pipelines hold the same scenes here, so
    # their records must be byte-identical to each other and to a single-stream step's (checked below; `ok`).  The single-stream
    # loop of rounds 1-3 is timed right after it with the same protocol (value_single_stream); --single-stream makes it the headline.
    two = None
    dt_single = None
    pair = None
    if not args.single_stream:
        step.run()
        torch.cuda.synchronize()
        ref_rec = step.pipe.results.clone()
        ns = max(2, int(args.streams))
        others = [Step(args, rank, world, dev) for _ in range(ns - 1)]
        for o_ in others:
            o_.prepare_events()
        step_b = others[0]
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        pair = ([step] + others, streams)
        for s_, st_ in zip(*pair):
            st_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st_):
                for _ in range(max(2, args.warmup)):
                    s_.run()
        torch.cuda.synchronize()
        dt, evs = timed_loop(args.steps, pair=pair)
        intervals = completion_intervals(step_ends[id(pair)], group=len(pair[1]))
        same = bool(all(torch.equal(s_.pipe.results, ref_rec) for s_ in pair[0]))
        dt_single, evs_single = timed_loop(args.steps)
        intervals_single = completion_intervals(step_ends[id(None)])
        two = {"streams": ns, "records_identical_to_single_stream": same,
               "note": "steps alternate between two HIP streams with double-buffered state; records of both pipelines compared byte "
                       "for byte with a single-stream step's in this run; per-stage times of the headline are measured on the stage's "
                       "own stream while the other stream's kernels share the chip (per_stage_ms_single_stream: the same stages alone)"}
    else:
        dt, evs = timed_loop(args.steps)
        intervals = completion_intervals(step_ends[id(None)])
        intervals_single = None
        evs_single = evs
    # the other placement of the scale head (see --eager-scale-head), measured the same way right after the headline
    # loop (untimed for the headline): the reference's forward order when the headline uses the kept-pairs-only order
    # (the comparison loops below are one-rank diagnostics: a multi-GPU run times the headline loop only)
    dt_other = None
    if not args.no_reference_order and world == 1:
        step.eager = not step.eager
        step.run()
        dt_other, _ = timed_loop(args.steps, sample=False)
        step.eager = not step.eager

    # the same step with the tuple MLP on the f32-input matrix cores (the arithmetic of rounds 1-2), same loop protocol
    dt_native = None
    headline_arith = _models.MLP_ARITH
    if _models.MLP_ARITH == "split" and not args.no_native_arith and world == 1:
        _models.MLP_ARITH = "native"
        step.run()
        dt_native, _ = timed_loop(args.steps, sample=False)
        _models.MLP_ARITH = "split"

    # the same step with the MLP in f16x2 arithmetic (fp16 operand pairs, three products per K step: half the matrix-core work;
    # error against float64 reported under mlp_error_vs_f64.split_f16x2); same loop protocol; not the headline
    dt_f16 = None
    f16_agreement = None
    if _models.MLP_ARITH == "split" and not args.no_f16x2 and world == 1:
        _models.MLP_ARITH = "split16"
        step.run()
        dt_f16, _ = timed_loop(args.steps, sample=False)
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

    if os.environ.get("CPPF_BENCH_PER_STEP") and rank == 0:
        for i_, ev in enumerate(evs):
            row = {n1: round(e0.elapsed_time(e1), 3) for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:])}
            print("step %d: total %.3f  %s" % (i_, ev[0][1].elapsed_time(ev[-1][1]), {k_: v_ for k_, v_ in row.items() if v_ > 0.2}),
                  file=sys.stderr)
    # per-stage HIP-event times (ms per launch, averaged over the timed steps) on the stream the kernels ran on
    stage_ms = {}
    for ev in evs:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            stage_ms[n1] = stage_ms.get(n1, 0.0) + e0.elapsed_time(e1) / len(evs)
    stage_ms_single = {}
    for ev in evs_single:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            stage_ms_single[n1] = stage_ms_single.get(n1, 0.0) + e0.elapsed_time(e1) / len(evs_single)
    step_times = sorted(ev[0][1].elapsed_time(ev[-1][1]) for ev in evs)
    # With two streams a stage's HIP-event time on its own stream includes the time its kernels wait for the other stream's (two
    # persistent matrix-core kernels do not fit a CU together): the kernels' own durations -- what the roofline divides by -- are
    # the stage times of the single-stream loop of the same run (rocprofv3 shows the same durations in both modes,
    # profiles/r4_two_stream_trace.txt).  stage_ms_2s keeps the two-stream stage times for the record.
    stage_ms_2s = stage_ms
    if dt_single is not None:
        stage_ms = stage_ms_single

    failed = False
    if rank == 0:
        B, N, T, R, S = step.B, step.N, step.T, args.rots, step.pipe.S
        res = step.pipe.results_to_numpy()
        all_rec = step.pipe.results_to_numpy(step.all_records)
        assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), res.tobytes())
        G = int(np.mean(res["ncell"]))
        mlp_stages = ("shot_encoder", "tuple_mlp", "scale_head")
        hip_stages = [s for s in Step.STAGES if s not in mlp_stages and s != "gather"]
        shared = set()      # every stage runs alone on the one stream (nothing shares the chip with another stage)
        # the HBM roofline object describes the longest of the kernels that ARE bandwidth-bound (section 4 of DESIGN.md); the
        # voting and descriptor kernels (VALU / LDS bound) have their fractions in per_kernel
        hbm_bound = ("decode_bins", "encode_tuples", "sample_tuples")
        dominant = max([s for s in hbm_bound if s not in shared], key=lambda s: stage_ms.get(s, 0.0))
        rows = []
        per_kernel = {}
        for s in Step.STAGES:
            ms = stage_ms.get(s, 0.0)
            ab = algorithmic_bytes(s, B, N, T, R, S, G)
            gbs = (ab / 1e9) / (ms / 1e3) if ms > 0 and ab else 0.0
            rows.append((s, ms, ab / 1e6, gbs))
            if s in hip_stages:
                # every stage with the bound that limits ITS kernel: the streaming kernels against the HBM peak
                # (algorithmic bytes / time; the counters' bytes beside them), the voting and descriptor kernels by the busy
                # fraction of the unit they saturate (VALU or LDS, from this run's SQ counter pass) with their work in the
                # domain's units (votes, cone tests, points) per second; the latency-bound ones (one workgroup per scene,
                # dependent phases) carry no fraction
                tr = pmc_traffic(s)
                act = unit_activity(counter_entry(STAGE_KERNEL.get(s, "")))
                e = dict(kernel=STAGE_KERNEL.get(s), ms=round(ms, 4), alg_MB=round(ab / 1e6, 2),
                         pmc_MB=None if tr is None else round(tr / 1e6, 2), activity=act)
                kind = STAGE_BOUND.get(s, "latency")
                if kind == "hbm":
                    e.update(bound="hbm", alg_GBs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4),
                             pmc_frac=None if (tr is None or ms <= 0) else round(tr / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4))
                elif kind == "unit":
                    if act is not None:
                        unit = "valu" if act["valu_busy"] >= act["lds_busy"] else "lds"
                        e.update(bound=unit, frac=act[unit + "_busy"], frac_kind="busy cycles of the %s pipe / shader cycles of the launch "
                                 "(SQ counters of this run)" % unit.upper())
                    else:
                        e.update(bound="valu", frac=None, frac_kind="no SQ counter pass in this run (%s)" % COUNTERS.get("reason", "--no-counters"))
                    sec = ms / 1e3 if ms > 0 else float("nan")
                    if s == "vote_center":
                        e["work"] = {"votes_per_launch": B * T * R, "votes_per_s": B * T * R / sec}
                    elif s == "rot_bins":
                        tf = T // 10
                        e["work"] = {"candidates_per_launch": 2 * B * tf * R, "candidates_per_s": 2 * B * tf * R / sec,
                                     "exhaustive_equivalent_compare_accumulates_per_s": 2.0 * B * tf * R * S / sec,
                                     "note": "the lookup table tests <= 8 bins per candidate (0.46 on average) where the "
                                             "reference's mm tests all %d" % S}
                    elif s in ("shot_frames", "shot352"):
                        e["work"] = {"points_per_launch": B * N, "points_per_s": B * N / sec}
                else:
                    e.update(bound="latency", frac=None, frac_kind="one workgroup per scene / dependent phases: neither a "
                                                                   "bandwidth nor an issue bound applies")
                per_kernel[s] = e
        dom_ms = stage_ms[dominant]
        dom_bytes = algorithmic_bytes(dominant, B, N, T, R, S, G)
        achieved = (dom_bytes / 1e9) / (dom_ms / 1e3)
        hip_only_ms = sum(stage_ms.get(s, 0.0) for s in hip_stages)
        path_bytes = sum(algorithmic_bytes(s, B, N, T, R, S, G) for s in hip_stages)
        step_ms = 1e3 * (dt_single if dt_single is not None else dt) / args.steps      # the step the stage times belong to
        roofline = dict(bound="hbm", kernel=dominant, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=pmc_traffic(dominant), launch_ms=dom_ms,
                        kernel_name=STAGE_KERNEL.get(dominant),
                        algorithmic_bytes_per_launch=dom_bytes,
                        algorithmic_model=("compulsory bytes of the stage as it runs here x %d scenes per launch (bench.py:"
                                           "algorithmic_bytes): SURVEY.md 8d's per-scene figures, except the tuple encode in its "
                                           "gathered form (indices + points + normals in, 40 pair features + 5 global indices per "
                                           "tuple out: 4.1 MB per scene instead of 8d's 30.35 MB of rows) and the decode behind "
                                           "the fused bin draw (bins in, vote parameters out)" % B),
                        traffic_source=("rocprofv3 counter passes of this run (2 x FETCH_SIZE + WRITE_SIZE, separate passes, fresh "
                                        "child processes)" if "reason" not in COUNTERS else "null: " + str(COUNTERS.get("reason"))),
                        # the whole path's algorithmic bytes (HIP stages; ~54 MB/scene) over the whole step
                        # (MLP included) and over the HIP stages alone, as fractions of the HBM peak
                        pipeline_bytes_per_step=path_bytes,
                        pipeline_frac=(path_bytes / 1e9) / (step_ms / 1e3) / HBM_PEAK_GBS,
                        hip_only_ms=hip_only_ms,
                        hip_only_frac=(path_bytes / 1e9) / (hip_only_ms / 1e3) / HBM_PEAK_GBS if hip_only_ms > 0 else None,
                        hip_only_scenes_per_s=B * world / (hip_only_ms / 1e3) if hip_only_ms > 0 else None,
                        mlp_ms=sum(stage_ms.get(s, 0.0) for s in mlp_stages),
                        stages_sharing_the_chip_with_torch=sorted(shared),
                        per_kernel=per_kernel,
                        per_stage_ms={s: round(stage_ms.get(s, 0.0), 4) for s in Step.STAGES},
                        per_stage_ms_source=("the single-stream loop of this run (kernel durations; with two streams a stage's "
                                             "event time also counts the waits for the other stream's kernels: "
                                             "per_stage_ms_two_streams)" if dt_single is not None else "the headline loop"),
                        per_stage_ms_two_streams=({s: round(stage_ms_2s.get(s, 0.0), 4) for s in Step.STAGES}
                                                  if dt_single is not None else None))
        if _models.MLP_ARITH in ("split", "split16"):
            nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
            # The dominant kernel of the step is the tuple MLP (cppf_reslayer_split, 3 launches back to back: the stage
            # time is their sum): matrix-core bound.  `achieved` = the bf16 MFMA work it executes (6 exact-product MFMAs per
            # float32 product, K padded to 16) over the stage's HIP-event time, against the dense bf16 peak; the
            # float32-equivalent rate (2 M K N of the layers) is next to it.  The HBM-bound kernel's roofline stays under "hbm".
            def layer_flops(k, n, proj):
                return 2.0 * n * ((k + 15) // 16 * 16) * (2 if proj else 1) + 2.0 * n * n, 2.0 * n * k * (2 if proj else 1) + 2.0 * n * n
            layers = [(360, 128, True)] + [(128, 128, False)] * 4 + [(128, 256, True), (256, 256, False), (256, 256, False), (256, 192, True)]
            if args.eager_scale_head:           # the scale head's matrix-core layers run inside this stage too (on every tuple)
                layers += [(256, 128, True), (128, 64, True)]
            executed = nprod * sum(layer_flops(*l)[0] for l in layers) * B * T
            algorithmic = sum(layer_flops(*l)[1] for l in layers) * B * T
            mlp_ms_ = stage_ms["tuple_mlp"]
            hbm = {k_: roofline[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms",
                                               "kernel_name", "algorithmic_bytes_per_launch", "algorithmic_model", "traffic_source")}
            roofline.update(bound="mfma", kernel="tuple_mlp", kernel_name="reslayer_split_kernel",
                            achieved=executed / 1e12 / (mlp_ms_ / 1e3), peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=executed / 1e12 / (mlp_ms_ / 1e3) / BF16_MFMA_PEAK_TFLOPS, traffic=pmc_traffic_mlp(3 if nprod == 6.0 else 2),
                            launch_ms=mlp_ms_, launches=3,
                            frac_kind="executed_bf16_mfma" if nprod == 6.0 else "executed_fp16_mfma",
                            traffic_covers="the same 3 launches as launch_ms (counter passes of this run: 2 x FETCH_SIZE + WRITE_SIZE)",
                            mfma_busy_per_launch={k_: (unit_activity(COUNTERS.get(k_.replace(", 3, 0>", ", %d, 0>" % (3 if nprod == 6.0 else 2)))
                                                                     or COUNTERS.get(k_.replace("#large", "").replace(", 3, 0>", ", %d, 0>" % (3 if nprod == 6.0 else 2)))) or {})
                                                  for k_ in TUPLE_MLP_KERNELS},
                            executed_bf16_flops_per_step=executed, algorithmic_f32_flops_per_step=algorithmic,
                            algorithmic_f32_tflops=algorithmic / 1e12 / (mlp_ms_ / 1e3), f32_input_mfma_peak_tflops=F32_MFMA_PEAK_TFLOPS,
                            # the same launch time against the other two readings of "algorithmic / peak"
                            frac_algorithmic_of_f32_input_mfma_peak=algorithmic / 1e12 / (mlp_ms_ / 1e3) / F32_MFMA_PEAK_TFLOPS,
                            frac_algorithmic_of_bf16_peak=algorithmic / 1e12 / (mlp_ms_ / 1e3) / BF16_MFMA_PEAK_TFLOPS,
                            algorithmic_model="tuple MLP of train_shot.py:48-73 at %d tuples: 2 M K N per Linear; executed = %d %s "
                                              "MFMA products per float32 product (%s)"
                                              % (B * T, int(nprod), "bf16" if nprod == 6.0 else "fp16",
                                                 "3-way exact operand split" if nprod == 6.0 else "fp16 operand pairs, 22-23 bits"),
                            hbm=hbm)
            roofline.pop("algorithmic_bytes_per_launch", None)
        # sanity of the synthetic workload: pose agreement with ground truth (5 deg / 5 cm on the up axis + centre)
        ok = 0
        for b in range(B):
            sc = step.scenes[b]
            terr = np.linalg.norm(res["t"][b] - sc["t"])
            cosang = abs(float(res["R"][b][:, 1] @ sc["R"][:, 1]))
            if terr < 0.05 and np.degrees(np.arccos(min(cosang, 1.0))) < 5.0:
                ok += 1
        cpu = None
        agree = None
        if args.cpu_scenes > 0 and world == 1:     # CPU baseline: rank 0 at N=1 only
            cpu, outs = cpu_baseline(args, step)
            from cppf2_amd.metrics import rt_degree_cm

            def rt(R, t):
                m = np.eye(4)
                m[:3, :3], m[:3, 3] = R, t
                return m
            errs = [rt_degree_cm(rt(res["R"][b], res["t"][b]), rt(o["R_est"], o["T_est"]), "bottle", clip=True)
                    for b, o in enumerate(outs)]
            agree = dict(scenes=len(outs), match_5deg5cm=float(np.mean([e[0] <= 5 and e[1] <= 5 for e in errs])),
                         max_rot_err_deg=float(max(e[0] for e in errs)), max_shift_cm=float(max(e[1] for e in errs)),
                         centre_argmax_equal=int(sum(int(res["argmax"][b]) == o["argmax"] for b, o in enumerate(outs))),
                         up_bin_equal=int(sum(int(res["up_idx"][b]) == o["up_idx"] for b, o in enumerate(outs))),
                         right_bin_equal=int(sum(int(res["right_idx"][b]) == o["right_idx"] for b, o in enumerate(outs))))
        evidence = arithmetic_evidence(step) if world == 1 and not args.no_evidence else {}
        total_scenes = B * world * args.steps
        line = {
            "metric": "scenes/sec (1/2/4/8 GPU) at 4096 pts x 20k tuples; 5deg5cm match vs ref",
            "value": total_scenes / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SHOT model, %d scenes/GPU x %d pts x %d tuples x %d rots, "
                                   "720 sphere bins, res 2 mm, bottle axes; random-init weights + teacher prior; scale head on %s; "
                                   "%s; MLP arithmetic: %s; tuple rows %s"
                                   % (B, N, T, R, "all tuples" if args.eager_scale_head else "the kept pairs only",
                                      "every step on one HIP stream" if args.single_stream else
                                      "consecutive steps alternate between two HIP streams (double-buffered state)",
                                      "float32 operands split exactly into 3 x bf16, 6 exact products on the bf16 matrix cores, "
                                      "float32 accumulate (float32-equivalent accuracy, tests/test_mlp_split.py)"
                                      if _models.MLP_ARITH == "split" else
                                      ("float32 operands as fp16 pairs (22-23 significant bits), 3 products on the fp16 matrix "
                                       "cores, float32 accumulate (error vs float64 at the library float32 GEMMs' level; NOT exact "
                                       "products)" if _models.MLP_ARITH == "split16" else "f32-input matrix cores"),
                                      ("gathered inside the first ResLayer's kernel (never written)" if GATHERED_TUPLES
                                       else "materialised ([T, 360] float32)")
                                      + ("; bins drawn in the epilogue of the logit head's output layer (logits never written)"
                                         if FUSED_DRAW else "")),
                       "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "parallelism": "scene-sharded x%d" % world},
            # the same run with the scale head on every tuple (the reference's forward order), same loop protocol
            "value_reference_order" if not args.eager_scale_head else "value_kept_pairs_order":
                (total_scenes / dt_other) if dt_other else None,
            # the same run with the MLP on the f32-input matrix instruction (no operand splitting)
            "value_f32_input_mfma": (total_scenes / dt_native) if dt_native else None,
            # the same run with the MLP in f16x2 arithmetic (operands as fp16 pairs, 22-23 bits; not the headline: products are
            # not exact there -- its error against float64 is under mlp_error_vs_f64.split_f16x2)
            "value_f16x2_mfma": (total_scenes / dt_f16) if dt_f16 else None,
            "f16x2_agreement": f16_agreement,
            # the headline's stream mode; value_single_stream = the same steps on ONE stream (rounds 1-3), same loop protocol
            "two_streams": two,
            "value_single_stream": (total_scenes / dt_single) if dt_single else None,
            "ms_per_step_single_stream": (1e3 * dt_single / args.steps) if dt_single else None,
            # intervals between consecutive step completions inside the timed loop (one HIP event per step), and the same for the
            # single-stream loop: the per-step spread of the headline
            "step_interval_ms": intervals,
            "step_interval_ms_single_stream": intervals_single,
            # first-event to last-event time of the sampled steps on their own stream: a step's RESIDENCE on its stream (with two
            # streams a step overlaps its neighbours, so this is about two step intervals -- not a per-step time)
            "step_residence_ms_sampled": {"min": round(step_times[0], 4), "median": round(step_times[len(step_times) // 2], 4),
                                "max": round(step_times[-1], 4), "n": len(step_times)},
            "records_gathered": int(all_rec.shape[0]),
            # SHA-256 of the gathered records in global scene order: equal for every world size and stream mode (the records
            # depend on the global scene id only)
            "records_sha256": __import__("hashlib").sha256(all_rec.tobytes()).hexdigest(),
            "host_cores_of_rank0": affinity,
            # the path's one collective: all_gather of the 160-byte scene records, HIP-event time of the stage
            "collective": {"backend": backend or "none (one rank: the local records are the result)", "world": world,
                           "op": "all_gather_into_tensor" if backend == "nccl" else ("all_gather" if backend else None),
                           "records_gathered": int(all_rec.shape[0]), "bytes_per_rank": int(B * 160),
                           "gather_us": round(1e3 * stage_ms.get("gather", 0.0), 2)},
            "roofline": roofline, "cpu_baseline": cpu,
            "pose_5deg5cm_vs_gt": ok / B, "oracle_agreement": agree,
        }
        line.update(evidence)
        # self-checks of the run: a comparison loop whose records differ from the headline's is a failed run, not a footnote
        problems = []
        if two is not None and not two["records_identical_to_single_stream"]:
            problems.append("two_streams: records differ from the single-stream ones")
        if f16_agreement is not None and f16_agreement["scenes_with_equal_argmax_rotation_bins_kept_count"] != f16_agreement["scenes"]:
            problems.append("f16x2_agreement: a scene's arg-max / rotation bins / kept count differs from the headline arithmetic's")
        line["ok"] = not problems
        line["problems"] = problems
        if args.breakdown:
            print("%-22s %10s %12s %10s" % ("stage", "ms/launch", "alg MB", "GB/s"), file=sys.stderr)
            for r in rows:
                print("%-22s %10.3f %12.1f %10.1f" % r, file=sys.stderr)
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
