"""Batched voting pipeline: eval.py:207-313 for B scenes at once, one launch per stage.

The reference processes one object instance at a time with >= 6 host syncs per instance
(SURVEY.md section 3.1).  Here every stage takes B scenes (ragged through offset arrays), all
buffers are allocated once, nothing is copied to the host until the final per-scene record
(CppfSceneResult, 160 bytes) is read, and no stage depends on a host-side value computed from
device data (grid sizes are bounded by `cells_cap`, kept-pair counts by the percentile index).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import SceneResult

_L = _lib.load()

RESULT_DTYPE = np.dtype([("argmax", "<i8"), ("t", "<f8", (3,)), ("R", "<f8", (3, 3)), ("scale", "<f4", (3,)),
                         ("peak", "<u4"), ("up_idx", "<i4"), ("right_idx", "<i4"), ("kept", "<i4"),
                         ("up_count", "<f4"), ("right_count", "<f4"), ("flags", "<i4"), ("ncell", "<i4"), ("pad_", "<i4", (3,))])
assert RESULT_DTYPE.itemsize == C.sizeof(SceneResult) == 160


class VotingPipeline:
    """Decode -> centre vote -> back-vote filter -> rotation votes -> pose, for a fixed batch geometry.

    cfg_up / cfg_right / cfg_front are the YAML axis vectors (config/config.yaml:12-14).  The reference hands
    (up, front, right) to generate_target_pairs(point_pairs, up, right, front) (eval.py:237-240), so column 2
    of targets_rot is the angle to cfg_right; that quirk is reproduced here.
    """

    def __init__(self, counts_points, counts_tuples, k=5, res=2e-3, num_rots=180, angle_tol=1.0,
                 backproj_ratio=0.1, imp_wt_margin=0.01, bmm_size=ops.BMM_SIZE, cfg_up=(0, 1, 0), cfg_right=(1, 0, 0),
                 cfg_front=(0, 0, 1), cells_cap=1 << 21, vote_mode=0, sphere_pts=None, device=None, trig=None):
        self.dev = device or ops._dev()
        self.B = len(counts_points)
        assert len(counts_tuples) == self.B
        self.np_, self.nt_ = list(map(int, counts_points)), list(map(int, counts_tuples))
        self.pt_off = ops._offsets(self.np_, self.dev)
        self.tup_off = ops._offsets(self.nt_, self.dev)
        self.Ntot, self.Ttot = int(sum(self.np_)), int(sum(self.nt_))
        self.max_n, self.max_t = max(self.np_), max(self.nt_)
        self.k, self.res, self.R = int(k), float(res), int(num_rots)
        self.angle_tol, self.ratio, self.margin, self.bmm = float(angle_tol), float(backproj_ratio), float(imp_wt_margin), int(bmm_size)
        self.cfg_up, self.cfg_right, self.cfg_front = np.array(cfg_up), np.array(cfg_right), np.array(cfg_front)
        # positional (up, right, front) of generate_target_pairs <- (cfg.up, cfg.front, cfg.right), eval.py:237-240
        self.axes = ops._axes9(self.cfg_up, self.cfg_front, self.cfg_right)
        self.up_axis = int(np.nonzero(self.cfg_up)[0][0])
        self.right_axis = int(np.nonzero(self.cfg_right)[0][0])
        self.cells_cap = int(cells_cap)
        self.vote_mode = int(vote_mode)
        sph = ops.sphere_bins(angle_tol) if sphere_pts is None else np.asarray(sphere_pts, dtype=np.float32)
        self.sphere_np = sph
        self.S = sph.shape[0]
        self.sphere = torch.from_numpy(sph).to(self.dev)
        self.cos_thr = ops.cone_threshold(angle_tol)
        self.lut = ops.bin_lut_device(sph, self.cos_thr, self.dev)    # None -> exhaustive bin sweep
        self.cs, self.sn = ops._trig(self.R, trig, self.dev)
        kg = [ops.percentile_params(n, self.ratio) for n in self.nt_]
        self.kidx = torch.tensor([a for a, _ in kg], dtype=torch.int32, device=self.dev)
        self.gamma = torch.tensor([b for _, b in kg], dtype=torch.float32, device=self.dev)
        self.max_kept = max(a for a, _ in kg) + 1
        d, B, T = self.dev, self.B, self.Ttot
        e = torch.empty
        self.grids = e((B, 32), dtype=torch.uint8, device=d)
        self.bins = e((T, 6), dtype=torch.int32, device=d)
        self.scaled = e((T, 2, 3), dtype=torch.float32, device=d)
        self.scale = e((T,), dtype=torch.float32, device=d)
        self.tr = e((T, 2), dtype=torch.float32, device=d)
        self.rot = e((T, 3), dtype=torch.float32, device=d)
        self.argmax = e((B,), dtype=torch.int64, device=d)
        self.peak = e((B,), dtype=torch.int32, device=d)
        self.world = e((B, 3), dtype=torch.float64, device=d)
        self.mask = e((T,), dtype=torch.uint8, device=d)
        self.kept_tuple = e((T,), dtype=torch.int32, device=d)
        self.kept_count = e((B,), dtype=torch.int32, device=d)
        self.kept_wt = e((T,), dtype=torch.float64, device=d)
        self.kept_row0 = e((T,), dtype=torch.int32, device=d)
        self.errs = e((T,), dtype=torch.float32, device=d)
        self.thr = e((B,), dtype=torch.float32, device=d)
        self.counts = e((2, B, self.S), dtype=torch.float32, device=d)
        self.top_idx = e((2, B), dtype=torch.int32, device=d)
        self.top_cnt = e((2, B), dtype=torch.float32, device=d)
        # result records: two slots (one per model of the ensemble, eval.py:219) + the selected ones; `results` is the slot the
        # stages currently write (slot 0 unless use_slot() says otherwise)
        self.result_slots = e((2, B, 160), dtype=torch.uint8, device=d)
        self.results = self.result_slots[0]
        self._slot = 0
        self.selected = e((B, 160), dtype=torch.uint8, device=d)
        self.losses = e((2, B), dtype=torch.float64, device=d)
        self.best = e((B,), dtype=torch.float64, device=d)
        self.ws_vote_bytes = _L.cppf_vote_center_workspace_bytes(B, self.cells_cap, self.Ttot)
        self.ws_bv_bytes = _L.cppf_backvote_workspace_bytes(self.Ntot, B)
        self.ws_rot_bytes = _L.cppf_rot_bins_workspace_bytes(B, self.S, self.max_kept, self.R, self.bmm)
        self.nb = 32
        self.ws = e((max(self.ws_vote_bytes, self.ws_bv_bytes, self.ws_rot_bytes, 256),), dtype=torch.uint8, device=d)

    def twin(self):
        """A second pipeline of the same batch geometry with working buffers of its own (bins, vote parameters, kept lists,
        workspace) that shares THIS pipeline's result slots, losses and selection buffers: the two model passes of the ensemble
        (eval.py:219) -- or two consecutive batches -- can then run on two HIP streams at once, each pass writing its own
        record slot, and select() / the final read see both.  The caller orders the streams (the SHOT pass' alignment loss reads
        the DINO pass' scale: an event between assemble() on one stream and alignment_loss() on the other)."""
        t = VotingPipeline(self.np_, self.nt_, k=self.k, res=self.res, num_rots=self.R, angle_tol=self.angle_tol,
                           backproj_ratio=self.ratio, imp_wt_margin=self.margin, bmm_size=self.bmm, cfg_up=self.cfg_up,
                           cfg_right=self.cfg_right, cfg_front=self.cfg_front, cells_cap=self.cells_cap, vote_mode=self.vote_mode,
                           sphere_pts=self.sphere_np, device=self.dev, trig=(self.cs, self.sn))
        t.result_slots, t.selected, t.losses, t.best = self.result_slots, self.selected, self.losses, self.best
        t.results = t.result_slots[0]
        return t

    # -- stages ---------------------------------------------------------------------------------
    def decode(self, pts, idx, logits, uniforms, prior=None):
        """prior (optional, same shape as logits) is added to the logits inside the kernel (== logits + prior)."""
        self.nb = int(logits.shape[-1])
        if prior is None:
            _lib.check(_L.cppf_decode_bins(self.B, ops._p(logits), logits.shape[-1], ops._p(uniforms), ops._p(pts),
                                           ops._p(idx), self.k, ops._p(self.pt_off), ops._p(self.tup_off), self.Ttot,
                                           self.axes, ops._p(self.bins), ops._p(self.scaled), ops._p(self.scale),
                                           ops._p(self.tr), ops._p(self.rot), ops._stream()), "cppf_decode_bins")
        else:                      # the synthetic benchmarks' teacher (the reference has no prior: cppf_hip_experimental.h)
            _lib.check(_L.cppf_decode_bins_prior(self.B, ops._p(logits), ops._p(prior), logits.shape[-1], ops._p(uniforms), ops._p(pts),
                                                 ops._p(idx), self.k, ops._p(self.pt_off), ops._p(self.tup_off), self.Ttot,
                                                 self.axes, ops._p(self.bins), ops._p(self.scaled), ops._p(self.scale),
                                                 ops._p(self.tr), ops._p(self.rot), ops._stream()), "cppf_decode_bins_prior")

    def decode_from_bins(self, pts, idx, nb=32):
        """The vote parameters from bins already drawn into self.bins (by the MLP's output layer:
        ops.reslayer_split_decode(..., bins=pipe.bins)): the second half of decode()."""
        self.nb = int(nb)
        _lib.check(_L.cppf_decode_from_bins(self.B, ops._p(self.bins), self.nb, ops._p(pts), ops._p(idx), self.k,
                                            ops._p(self.pt_off), ops._p(self.tup_off), self.Ttot, self.axes,
                                            ops._p(self.scaled), ops._p(self.scale), ops._p(self.tr), ops._p(self.rot),
                                            ops._stream()), "cppf_decode_from_bins")

    def vote_center(self, pts, idx, grid=None, grid_off=None, vote_wt=None, phase=0):
        """phase 0: bounds + frames + votes + argmax in one go.  phase 1: bounds + per-pair frames only;
        phase 2: votes + argmax on the frames left by phase 1 (CPPF_VC_FRAMES_ONLY / _READY, LDS-slab mode)."""
        st = ops._stream()
        mode = self.vote_mode
        if phase:
            mode = 1 | (0x200 if phase == 1 else 0x100)
        if phase != 2:
            _lib.check(_L.cppf_scene_bounds(self.B, ops._p(pts), ops._p(self.pt_off), C.c_float(self.res),
                                            ops._p(self.grids), st), "cppf_scene_bounds")
        _lib.check(_L.cppf_vote_center(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k,
                                       ops._p(self.tup_off), self.max_t, self.Ttot, ops._p(self.tr), ops._p(vote_wt),
                                       C.c_double(self.res),
                                       self.R, ops._p(self.cs), ops._p(self.sn), ops._p(self.grids), ops._p(grid),
                                       ops._p(grid_off), self.cells_cap, mode, ops._p(self.ws),
                                       self.ws_vote_bytes, ops._p(self.argmax), ops._p(self.peak), ops._p(self.world),
                                       st), "cppf_vote_center")

    def backvote(self, pts, idx):
        _lib.check(_L.cppf_backvote_filter(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k,
                                           ops._p(self.tup_off), ops._p(self.tr), ops._p(self.world), self.axes,
                                           ops._p(self.kidx), ops._p(self.gamma), C.c_double(self.margin), self.R,
                                           ops._p(self.mask), ops._p(self.kept_tuple), ops._p(self.kept_count),
                                           ops._p(self.kept_wt), ops._p(self.kept_row0), ops._p(self.errs),
                                           ops._p(self.thr), ops._p(self.ws), self.ws_bv_bytes, ops._stream()),
                   "cppf_backvote_filter")

    def rot_bins(self, pts, idx, use_lut=True):
        """Both rotation votes (eval.py:277-293; angle columns 0 and 2 of targets_rot) in one pass over the kept pairs."""
        lut = self.lut if use_lut else None
        _lib.check(_L.cppf_rot_bins2(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k,
                                     ops._p(self.tup_off), ops._p(self.rot), 0, 2, ops._p(self.kept_tuple),
                                     ops._p(self.kept_count), ops._p(self.kept_wt), ops._p(self.kept_row0),
                                     self.max_kept, self.R, ops._p(self.cs), ops._p(self.sn), ops._p(self.sphere),
                                     self.S, C.c_float(self.cos_thr), self.bmm, ops._p(lut), ops.LUT_ROWS,
                                     ops.LUT_COLS, ops._p(self.counts), ops._p(self.top_idx), ops._p(self.top_cnt),
                                     ops._p(self.ws), self.ws_rot_bytes, ops._stream()), "cppf_rot_bins2")

    def rot_bins_single(self, pts, idx, axis, use_lut=True):
        """One rotation vote through the single-axis entry point (axis 0 = up, 1 = right); same results."""
        lut = self.lut if use_lut else None
        _lib.check(_L.cppf_rot_bins(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k,
                                    ops._p(self.tup_off), ops._p(self.rot), (0, 2)[axis], ops._p(self.kept_tuple),
                                    ops._p(self.kept_count), ops._p(self.kept_wt), ops._p(self.kept_row0),
                                    self.max_kept, self.R, ops._p(self.cs), ops._p(self.sn), ops._p(self.sphere),
                                    self.S, C.c_float(self.cos_thr), self.bmm, ops._p(lut), ops.LUT_ROWS,
                                    ops.LUT_COLS, ops._p(self.counts[axis]), ops._p(self.top_idx[axis]),
                                    ops._p(self.top_cnt[axis]), ops._p(self.ws), self.ws_rot_bytes, ops._stream()),
                   "cppf_rot_bins")

    def kept_rows(self):
        """Global tuple rows of the pairs that survived the back-vote filter, int64 [B * max_kept], without a host sync:
        scene b's list is padded to max_kept entries by repeating its first row (harmless for gather / scatter)."""
        rows = torch.empty((self.B * self.max_kept,), dtype=torch.int64, device=self.dev)
        _lib.check(_L.cppf_kept_rows(self.B, ops._p(self.tup_off), ops._p(self.kept_tuple), ops._p(self.kept_count),
                                     self.max_kept, ops._p(rows), ops._stream()), "cppf_kept_rows")
        return rows

    def kept_rows32(self):
        """kept_rows() as int32 [B * max_kept] in a buffer allocated once (what the gathering MLP kernel and
        ops.reslayer_tail index with); a scene's list is padded to max_kept entries with a valid row that is never written."""
        if getattr(self, "_rows32", None) is None:
            self._rows32 = torch.empty((self.B * self.max_kept,), dtype=torch.int32, device=self.dev)
        _lib.check(_L.cppf_kept_rows32(self.B, ops._p(self.tup_off), ops._p(self.kept_tuple), ops._p(self.kept_count),
                                       self.max_kept, ops._p(self._rows32), ops._stream()), "cppf_kept_rows32")
        return self._rows32

    def scatter_kept(self, rows, values, out=None):
        """[T, C] buffer holding `values` at `rows` (what assemble() reads for the kept pairs); other rows are untouched.
        Only the real entries of each scene's list are written (the padded ones repeat a row: an index_put with duplicate
        indices is order-dependent unless the duplicates carry identical values)."""
        if out is None:
            out = torch.zeros((self.Ttot, values.shape[1]), dtype=values.dtype, device=self.dev)
        keep = (torch.arange(self.max_kept, device=self.dev)[None, :] < self.kept_count[:, None]).reshape(-1)
        out[rows.long()[keep]] = values[keep]
        return out

    def pose_tensors(self):
        """Views into the device result records: t float64 [B,3], R float64 [B,3,3], scale float32 [B,3], flags int32 [B]."""
        r64 = self.results.view(torch.float64)                       # 160 B = 20 doubles per record
        return (r64[:, 1:4], r64[:, 4:13].reshape(self.B, 3, 3), self.results[:, 104:116].view(torch.float32),
                self.results[:, 140:144].view(torch.int32).reshape(self.B))

    def use_slot(self, model_idx):
        """The record slot (0 = DINO pass, 1 = SHOT pass of the ensemble) that assemble() / refine() write from now on."""
        self._slot = int(model_idx)
        self.results = self.result_slots[self._slot]
        return self.results

    def alignment_loss(self, pts, idx, y_only, model_idx=None, scale_slot=0):
        """eval.py:358-363 for every scene on the device (cppf_alignment_loss): mean over the kept pairs of the clipped L1
        distance between their canonicalised points, (pc - T_est) @ R_est / pred_scale_norm in float64, and the decoded
        (un-scaled) coordinates bins / 31 - 0.5; the y coordinate only for the up-symmetric categories.  The scale norm is
        that of slot `scale_slot`'s records (the DINO pass' for both passes, eval.py:308-310).  Scores the records of the
        current slot (or slot model_idx) into self.losses[slot]; returns that float64 [B] view (NaN if nothing was kept)."""
        slot = self._slot if model_idx is None else int(model_idx)
        _lib.check(_L.cppf_alignment_loss(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k, ops._p(self.tup_off),
                                          ops._p(self.bins), self.nb, ops._p(self.kept_tuple), ops._p(self.kept_count),
                                          int(bool(y_only)), ops._p(self.result_slots[slot]), ops._p(self.result_slots[scale_slot]),
                                          ops._p(self.losses[slot]), ops._stream()), "cppf_alignment_loss")
        return self.losses[slot]

    def select(self, enable0=True, enable1=True):
        """eval.py:365-372 on the device (cppf_ensemble_select): self.selected[b] = the record of the model with the smaller
        alignment loss (strict '<' against inf, the DINO pass first; enable0 = geo_branch, enable1 = visual_branch), carrying
        the DINO pass' scale and the pick in pad_[0]; self.best = the winning losses."""
        _lib.check(_L.cppf_ensemble_select(self.B, ops._p(self.result_slots[0]), ops._p(self.result_slots[1]),
                                           ops._p(self.losses[0]), ops._p(self.losses[1]), int(bool(enable0)), int(bool(enable1)),
                                           ops._p(self.selected), ops._p(self.best), ops._stream()), "cppf_ensemble_select")
        return self.selected

    def assemble(self, pred_scales=None):
        _lib.check(_L.cppf_assemble_pose(self.B, ops._p(self.sphere), ops._p(self.top_idx[0]), ops._p(self.top_cnt[0]),
                                         ops._p(self.top_idx[1]), ops._p(self.top_cnt[1]), self.up_axis,
                                         self.right_axis, ops._p(self.argmax), ops._p(self.peak), ops._p(self.world),
                                         ops._p(self.grids), ops._p(pred_scales), ops._p(self.tup_off),
                                         ops._p(self.kept_tuple), ops._p(self.kept_count), ops._p(self.results),
                                         ops._stream()), "cppf_assemble_pose")

    def refine(self, pts, idx, y_only, steps=100, lr=1e-2):
        """Online alignment refinement of the assembled poses, in place (eval.py:319-355; `opt=True`)."""
        _lib.check(_L.cppf_refine_pose(self.B, ops._p(pts), ops._p(self.pt_off), ops._p(idx), self.k,
                                       ops._p(self.tup_off), ops._p(self.scaled), ops._p(self.kept_tuple),
                                       ops._p(self.kept_count), int(bool(y_only)), int(steps), C.c_float(lr),
                                       ops._p(self.results), ops._stream()), "cppf_refine_pose")

    def vote(self, pts, idx, logits, uniforms, pred_scales=None, grid=None, grid_off=None):
        """Everything after the MLP: eval.py:225-313.  All arguments are device tensors in the batch layout.
        Returns the device tensor of B result records (uint8 [B,160]); use results_to_numpy() to read them.
        logits=None: the bins are already in self.bins (drawn by the MLP's output layer, ops.reslayer_split_decode).
        pred_scales: float32 [T, 3], or a callable evaluated after the back-vote filter that returns it -- the scale head is read
        only for the kept pairs (eval.py:272), so a caller can run it on just those rows (kept_rows32 / kept_count / max_kept)."""
        if logits is None:
            self.decode_from_bins(pts, idx)
        else:
            self.decode(pts, idx, logits, uniforms)
        self.vote_center(pts, idx, grid, grid_off)
        self.backvote(pts, idx)
        self.rot_bins(pts, idx)
        self.assemble(pred_scales() if callable(pred_scales) else pred_scales)
        return self.results

    def capture(self, pts, idx, logits, uniforms, pred_scales=None):
        """Captures decode -> ... -> pose into one HIP graph (torch.cuda.CUDAGraph) bound to the given (static) input
        tensors; returns a callable that replays it.  The stages allocate nothing and never sync, so the whole
        post-MLP path becomes a single graph launch -- what matters at B = 1 (the reference's one-instance-at-a-time
        use), where ~15 kernel launches of a few microseconds each are otherwise host-bound."""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.vote(pts, idx, logits, uniforms, pred_scales)          # warm-up outside capture (attribute calls)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.vote(pts, idx, logits, uniforms, pred_scales)
        self._graph = g

        def replay():
            g.replay()
            return self.results
        return replay

    def results_to_numpy(self, results=None):
        r = self.results if results is None else results
        return np.frombuffer(r.cpu().numpy().tobytes(), dtype=RESULT_DTYPE).copy()


class BatchMode:
    """The batch mode of the path: consecutive, independent batches alternate between HIP streams, each stream with state of its
    own (one VotingPipeline / model pass object per stream: `states`), so that one batch's descriptor, voting and small MLP
    kernels run beside the other batch's wide matrix-core launches.  While the mode is active those persistent launches leave one
    CU per shader engine to the other streams (cppf_mlp_reserve_cus: workgroups are placed round-robin over the engines, and an
    engine without a free CU stalls the other stream's whole launch).  Results are those of the same batches run one after the
    other on one stream (every entry point takes its stream; no state is shared).

        with BatchMode([state_a, state_b]) as mode:
            for batch in batches:
                with mode.next() as state:        # the stream of this batch is current inside the block
                    state.run(batch)
        # leaving the block: the calling stream waits for both streams, the reservation is lifted
    """

    def __init__(self, states, device=None, reserve_cus=None, streams=None):
        self.states = list(states)
        assert len(self.states) >= 1
        self.dev = torch.device(device) if device is not None else ops._dev()
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(device=self.dev) for _ in self.states]
        assert len(self.streams) == len(self.states)
        self.reserve_cus = ops.batch_mode_reserved_cus(self.dev) if reserve_cus is None else int(reserve_cus)
        self.count = 0
        self.active = False

    def __enter__(self):
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            st.wait_stream(cur)
        self.prev_reserved = ops.mlp_reserve_cus(self.reserve_cus if len(self.states) > 1 else 0)
        self.active = True
        return self

    def __exit__(self, *exc):
        ops.mlp_reserve_cus(self.prev_reserved)     # (the enclosing block's reservation, not 0: ADVICE r5)
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)
        self.active = False
        return False

    def next(self):
        """Context manager: the next stream's state, with that stream current."""
        assert self.active, "use inside `with BatchMode(...)`"
        j = self.count % len(self.states)
        self.count += 1
        return _OnStream(self.streams[j], self.states[j])


class _OnStream:
    def __init__(self, stream, state):
        self.stream, self.state = stream, state
        self.ctx = None

    def __enter__(self):
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self.state

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)
