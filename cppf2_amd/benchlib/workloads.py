"""The three workloads bench.py times, each as a Step object that holds the resident inputs and runs ONE pass of the path:
Step (BASELINE configs[1], the headline: SHOT model), EnsembleStep (configs[2]: both models per instance), DenseStep (configs[4]:
65 536 pairs, float16 table, weighted votes).  Every launch of a pass is a kernel of libcppf_hip.so."""
import time

import numpy as np
import torch


class Cfg:
    num_more = 3
    res = 2e-3
    up, right, front = [0, 1, 0], [1, 0, 0], [0, 0, 1]     # config/config.yaml:12-14


class Step:
    """Holds the resident inputs and runs one pass of the path."""

    STAGES = ["sample_tuples", "shot_frames", "shot352", "shot_encoder", "encode_tuples", "tuple_mlp",
              "decode_bins", "vote_frames", "vote_center", "backvote_filter", "rot_bins", "scale_head", "assemble_pose", "gather"]

    def __init__(self, args, rank, world, dev, cloud=None, scene_shift=0):
        """scene_shift: this Step holds global scenes [scene_shift + rank B, scene_shift + (rank + 1) B) -- the second pipeline of
        the two-stream mode holds a DIFFERENT batch than the first (an aliased buffer between the two would then change records)."""
        from cppf2_amd import dist as cdist
        from cppf2_amd import ops, synth
        from cppf2_amd.models import BeyondCPPFShot
        from cppf2_amd.pipeline import VotingPipeline
        self.ops, self.dist, self.args, self.rank, self.world, self.dev = ops, cdist, args, rank, world, dev
        assert cdist.shard(args.scenes_per_gpu * world, rank, world) == (rank * args.scenes_per_gpu, (rank + 1) * args.scenes_per_gpu)
        B, N, T = args.scenes_per_gpu, args.points, args.tuples
        self.B, self.N, self.T = B, N, T
        self.scene0 = scene_shift + rank * B
        # clouds: BASELINE's synthetic surface samples, or the same kind of object at the point density eval.py:185-201's 2 mm voxel
        # grid gives real inputs (synth.make_scene_voxel2mm: ~250 neighbours inside the 2 cm SHOT support instead of ~90)
        self.cloud = cloud or getattr(args, "cloud", "synthetic")
        make = synth.make_scene_voxel2mm if self.cloud == "voxel2mm" else synth.make_scene
        scenes = [make(args.seed, self.scene0 + b, N) for b in range(B)]
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
        # as its generator (ops.BinPrior: the fused bin draw evaluates -0.5 ((k - pos) / sigma)^2 in its epilogue, 24 bytes per tuple
        # instead of a 983 MB array per launch; --array-prior passes the array -- the same values bit for bit, same records)
        self.teacher = ops.BinPrior(pos.contiguous(), 1.0 / 0.6)
        self._prior_dense = None
        self.prior = self.prior_dense if getattr(args, "array_prior", False) else self.teacher
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
    def prior_dense(self):
        """The teacher prior as a [T, 6, 32] array (the unfused decode kernel and --array-prior read it), built on first use."""
        if self._prior_dense is None:
            self._prior_dense = self.teacher.dense(32)
        return self._prior_dense

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
            # round 5: the 40 pair features too are built by the first ResLayer's kernel (cppf_reslayer_split_encode): no launch, no
            # per-tuple array between the sampler and the tuple encoder (--separate-encode: the two-kernel form of rounds 3-4)
            if getattr(a, "separate_encode", False):
                heads, gidx = ops.encode_tuples_shot_heads(self.pts, idx, normal, pipe.pt_off, pipe.tup_off)
            else:
                heads, gidx = ops.TupleSource(self.pts, idx, normal, pipe.pt_off, pipe.tup_off), None
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
            pipe.decode(self.pts, idx, logits, u, prior=self.prior_dense)      # teacher prior added inside the decode kernel
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

    def __init__(self, args, rank, world, dev, scene_shift=0):
        super().__init__(args, rank, world, dev, scene_shift=scene_shift)
        from cppf2_amd.models import BeyondCPPFDino
        torch.manual_seed(args.seed + 1)
        self.dino = BeyondCPPFDino(Cfg()).to(dev).eval()
        g = torch.Generator(device="cpu").manual_seed(args.seed + 17 + rank + 1000003 * scene_shift)
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
            heads, gidx = ops.TupleSource(self.pts, idx, None, pipe.pt_off, pipe.tup_off), None
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
            heads2, gidx2 = ops.TupleSource(self.pts, idx, normal, pipe.pt_off, pipe.tup_off), None
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
        heads, gidx = ops.TupleSource(self.pts, idx, None, pipe.pt_off, pipe.tup_off), None
        self._mark("dino_encode")
        _, tf = fused_stack((self.dino.tuple_encoder, self.dino.logit_encoder), None, gather=(heads, gidx, tables, fold),
                            decode=(u, self.prior, pipe.bins))
        self._mark("dino_tuple_mlp")
        self._vote_pass("dino_", self.dino, tf, idx, self.scales_buf)
        # ---- model 1: SHOT (train_shot.py:75-83, 117-122; eval.py:223) -------------------------------------------
        pipe.use_slot(1)
        u = ops.philox_uniform(T, 6, a.seed, 2, ids, self.dev)
        heads, gidx = ops.TupleSource(self.pts, idx, normal, pipe.pt_off, pipe.tup_off), None
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

