#!/usr/bin/env python
"""Entry point with the reference's eval.py flag surface (eval.py:54-65):

    python eval.py --angle_tol=1. --imp_wt_margin=0.01 --backproj_ratio=.1 --num_pairs=50000 --num_rots=180 \
                   --opt=False --geo_branch=True --visual_branch=True \
                   [--data=synthetic --num_scenes=16 --categories=bottle,mug | --category=bottle] [--ckpt_dir=ckpts]

Like the reference (eval.py:84-101) it sets up one DINO model + one SHOT model + one cfg per category of the whitelist
(all six by default) and evaluates every object instance with both models, keeping the pose with the smaller
alignment loss (eval.py:219-372).  The per-instance loop of the reference (tuple sampling -> SHOT -> two models ->
decode -> centre vote -> back-vote filter -> rotation votes -> pose -> ensemble selection) runs here batched over all
instances of a category on the GPU through cppf2_amd, with no host round trip before the final 160-byte records.
What the image cannot provide is stated, not faked:
  * NOCS REAL275 images / SAR-Net masks / last.ckpt / DINOv2 weights are absent -> `--data=synthetic` (default)
    evaluates seeded synthetic instances (cppf2_amd.synth) with random-init weights (or the checkpoints found under
    `--ckpt_dir`, laid out like the reference's: <ckpt_dir>/{dino,shot}/<cat>-num_more-3/{.hydra/config.yaml,
    lightning_logs/version_0/checkpoints/last.ckpt}, eval.py:91-99) plus a teacher prior; `--data=depth` evaluates
    one depth+mask PNG pair (example_data layout) through backproject/downsample.
  * the Adam/lietorch refinement (eval.py:319-355, `opt`) runs as one HIP kernel per batch
    (cppf_refine_pose); lietorch is absent, so its semantics are restated from the published algorithm and pinned
    only by the oracle (parity unpinned).
Swapped flag names are kept: geo_branch gates model 0 (DINO), visual_branch gates model 1 (SHOT) (eval.py:367).
"""
import json
import os
import sys

import numpy as np
import torch

from cppf2_amd import geometry, ops, shot, synth
from cppf2_amd.config import load_checkpoint_config, load_config
from cppf2_amd.models import BeyondCPPFDino, BeyondCPPFShot, load_reference_checkpoint
from cppf2_amd.ops import get_topk_dir  # noqa: F401  (the reference defines it in this file, eval.py:37-51; demo.py and the notebook import it from here)
from cppf2_amd.pipeline import VotingPipeline

id2category = {1: "bottle", 2: "bowl", 3: "camera", 4: "can", 5: "laptop", 6: "mug"}     # dataset.py:29-37
category2id = {v: k for k, v in id2category.items()}
WHITELIST = ["can", "bowl", "laptop", "bottle", "camera", "mug"]                          # eval.py:78
UP_SYM = ("can", "bottle", "bowl")                                                        # eval.py:333,362


def _flag(v):
    if isinstance(v, str):
        if v.lower() in ("true", "false"):
            return v.lower() == "true"
        try:
            return float(v) if any(c in v for c in ".e") else int(v)
        except ValueError:
            return v
    return v


def load_custom(ckpt_shot=None, ckpt_dino=None, config_dir="config", device=None):
    """The instance-level setup of the reference's demo (config/custom.yaml: no category group): (cfg, dino, shot)."""
    dev = device or ops._dev()
    cfg = load_config(config_dir, "custom", [])
    dino_model = BeyondCPPFDino(cfg).to(dev).eval()
    shot_model = BeyondCPPFShot(cfg).to(dev).eval()
    if ckpt_dino:
        load_reference_checkpoint(dino_model, ckpt_dino)
    if ckpt_shot:
        load_reference_checkpoint(shot_model, ckpt_shot)
    return cfg, dino_model, shot_model


def load_category(cat_name, ckpt_dir=None, ckpt_shot=None, ckpt_dino=None, config_dir="config", device=None):
    """eval.py:87-101 for one category: (cfg, dino_model, shot_model).  With a checkpoint directory each model is built
    from the cfg saved next to its weights (`.hydra/config.yaml`) and the loop keeps the SHOT run's cfg (the last
    assignment, eval.py:101); otherwise config/ + category group and random-init weights."""
    dev = device or ops._dev()
    cfg = load_config(config_dir, "config", ["category=%s" % cat_name])
    cfgs = {"dino": cfg, "shot": cfg}
    weights = {"dino": ckpt_dino, "shot": ckpt_shot}
    if ckpt_dir:
        for name in ("dino", "shot"):
            root = os.path.join(ckpt_dir, name, "%s-num_more-3" % cat_name)               # eval.py:91,96
            hy = os.path.join(root, ".hydra", "config.yaml")
            if os.path.exists(hy):
                cfgs[name] = load_checkpoint_config(hy)                                    # eval.py:92,97
            ck = os.path.join(root, "lightning_logs", "version_0", "checkpoints", "last.ckpt")
            if weights[name] is None and os.path.exists(ck):
                weights[name] = ck
    dino_model = BeyondCPPFDino(cfgs["dino"]).to(dev).eval()
    shot_model = BeyondCPPFShot(cfgs["shot"]).to(dev).eval()
    if weights["dino"]:
        load_reference_checkpoint(dino_model, weights["dino"])
    if weights["shot"]:
        load_reference_checkpoint(shot_model, weights["shot"])
    return cfgs["shot"], dino_model, shot_model


def needed_cells(pc, res):
    """Cells of the vote grid of one instance (train_dino.py:173-175: int32 truncation of the float32 extent / res, + 1)."""
    ext = (pc.max(0) - pc.min(0)).astype(np.float32) / np.float32(res)
    return int(np.prod(ext.astype(np.int64) + 1))


_SIDE_STREAMS = {}


def _side_streams(dev):
    """The two HIP streams the model passes of run_ensemble run on (one pair per device for the life of the process: scratch
    buffers keyed by stream are reused from call to call)."""
    key = str(torch.device(dev))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    return _SIDE_STREAMS[key]


_PIPES = {}                # batch geometry -> (pipe, twin | None, [scales_buf, scales_buf], workspace bytes); least recently used first
PIPE_CACHE_MAX = 4
PIPE_CACHE_BYTES = 16 << 30        # bound on the cached pipelines' vote workspaces (B x cells_cap x 4 bytes each, twice with a twin)


def _pipelines(dev, Ns, num_pairs, k, cfg, num_rots, angle_tol, backproj_ratio, imp_wt_margin, cap, two):
    """The VotingPipeline (+ its twin for the two-stream mode, + the [T, 3] scale buffers of the two passes) of one batch geometry,
    kept from call to call: a streaming evaluation calls run_ensemble once per batch and category, and building a pipeline means
    ~25 device allocations, the workspace, and host-to-device copies of the offsets, sphere bins, bin lookup table and rotation
    table -- per call, before.  Keyed by everything the buffers' sizes and tables depend on; at most PIPE_CACHE_MAX geometries are
    kept, and at most PIPE_CACHE_BYTES of vote workspace (real batches are ragged: a batch with new point counts -- the usual case
    on REAL275 chunks -- builds its own; the oldest entries are dropped BEFORE the new one is built, so the peak is the bound, not
    the bound plus one).  The cache pays for repeated geometries: the synthetic mode, the benchmarks, fixed-size crops.  The
    buffers of a cached pipeline are overwritten by the next call with the same geometry: a caller that keeps run_ensemble's
    `pipe` reads it before."""
    key = (str(torch.device(dev)), tuple(Ns), int(num_pairs), int(k), float(cfg.res), int(num_rots), float(angle_tol),
           float(backproj_ratio), float(imp_wt_margin), tuple(cfg.up), tuple(cfg.right), tuple(cfg.front), int(cap), bool(two))
    hit = _PIPES.pop(key, None)
    if hit is None:
        cost = len(Ns) * int(cap) * 4 * (2 if two else 1)
        while _PIPES and (len(_PIPES) >= PIPE_CACHE_MAX or sum(v[3] for v in _PIPES.values()) + cost > PIPE_CACHE_BYTES):
            del _PIPES[next(iter(_PIPES))]           # evict first: the new pipeline never coexists with more than the bound
        pipe = VotingPipeline(Ns, [num_pairs] * len(Ns), k=k, res=cfg.res, num_rots=num_rots, angle_tol=angle_tol,
                              backproj_ratio=backproj_ratio, imp_wt_margin=imp_wt_margin, cfg_up=cfg.up,
                              cfg_right=cfg.right, cfg_front=cfg.front, cells_cap=cap)
        bufs = [torch.zeros((pipe.Ttot, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        hit = (pipe, pipe.twin() if two else None, bufs, cost)
    _PIPES[key] = hit
    return hit[:3]


@torch.no_grad()
def run_ensemble(cfg, dino_model, shot_model, pcs, descs, seed, scene_ids, num_pairs, num_rots, angle_tol=1.,
                 imp_wt_margin=0.01, backproj_ratio=.1, opt=False, geo_branch=True, visual_branch=True, up_sym=False,
                 priors=None, keep=False, scale_priors=None, two_streams=True):
    """eval.py:207-372 for a batch of instances of one category.  pcs: list of float32 [N_b,3]; descs: list of float32
    [N_b,1024] arrays or (device) tensors (DINOv2 features at the points: inputs to the path); priors: optional callable(idx_global, base) -> logit
    prior [T,6,nb] added to both models' logits; scale_priors: optional float32 [B,3] teacher box extents that stand in for
    the scale head of random-init weights (the head's output stays in the sum at 1e-3).  Returns dict(records=[2 x
    structured array], losses float64 [2,B], pick int [B], scale, scale_norm, idx, pipe, ...).
    two_streams (default): the DINO pass and the SHOT pass (descriptors included) run on two HIP streams at once, each with
    working buffers of its own (VotingPipeline.twin) -- one pass' voting and descriptor kernels beside the other's wide
    matrix-core kernels; the only cross-stream dependency is the DINO pass' scale, which scores the SHOT pass too
    (eval.py:308-310).  Same records as the one-stream order (keep=True, which hands out intermediates, uses that order)."""
    dev = ops._dev()
    B = len(pcs)
    Ns = [int(p.shape[0]) for p in pcs]
    k = cfg.num_more + 2
    for p in pcs:                                                                          # eval.py:200
        if ((p.max(0) - p.min(0)).max() / cfg.res) > 1000:
            raise ValueError("instance larger than 1000 cells: the reference skips it (eval.py:200); drop it from the batch")
    cap = max(1 << 18, max(needed_cells(p, cfg.res) for p in pcs))
    cap = 1 << int(np.ceil(np.log2(cap)))
    if cap * B > (1 << 33):
        raise ValueError("vote grids of %d cells x %d instances do not fit one batch; evaluate fewer instances per call" % (cap, B))
    pts = torch.from_numpy(np.concatenate(pcs)).to(dev)
    two = bool(two_streams) and not keep
    pipe, twin, scale_bufs = _pipelines(dev, Ns, num_pairs, k, cfg, num_rots, angle_tol, backproj_ratio, imp_wt_margin, cap, two)
    # eval.py:207 -- one tuple table per instance, shared by both models
    idx = torch.cat([ops.sample_tuples(n, num_pairs, k, seed, (s,), dev) for s, n in zip(scene_ids, Ns)])
    # descriptors: device tensors stay where they are (main_nocs samples them on the GPU), host arrays are uploaded one by one --
    # the batch is assembled on the device, not by a host-side copy of its largest input (16.8 MB per 4096 points)
    desc = torch.cat([d.to(dev) if torch.is_tensor(d) else torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32)).to(dev)
                      for d in descs])
    base = torch.cat([torch.full((num_pairs,), o, dtype=torch.int64) for o in np.cumsum([0] + Ns[:-1])]).to(dev)
    prior = priors(idx, base) if priors is not None else None      # a [T, 6, nb] array or an ops.BinPrior
    prior_arr = (lambda: prior.dense() if isinstance(prior, ops.BinPrior) else prior)
    scale_prior = None
    if scale_priors is not None:
        scale_prior = torch.from_numpy(np.asarray(scale_priors, dtype=np.float32)).to(dev).repeat_interleave(num_pairs, 0)
    main = torch.cuda.current_stream(dev)
    streams = _side_streams(dev) if two else [main, main]
    pipes = [pipe, twin if two else pipe]
    for st_ in streams:
        st_.wait_stream(main)
    kept = []
    extra = {}
    dino_scored = torch.cuda.Event() if two else None

    def one_pass(model_idx):
        model = (dino_model, shot_model)[model_idx]
        pp = pipes[model_idx]
        pp.use_slot(model_idx)                       # each pass writes its own records; nothing is read back before the end
        scales_buf = scale_bufs[model_idx]           # (rows of pairs that are not kept are never read: assemble() walks the kept list)
        u = torch.cat([ops.philox_uniform(num_pairs, 6, seed, 1 + model_idx, (s,), dev) for s in scene_ids])
        # eval.py:225-229 (the bin draw) runs as the epilogue of the logit head's output layer when the kernels allow it (split
        # arithmetic, no intermediates requested): the heads then return None in place of the logits.  The scale head is
        # evaluated after the back-vote filter, on the kept pairs' rows only (eval.py:272 reads nothing else).
        draw = None if keep else (u, None if prior is None else (prior if isinstance(prior, ops.BinPrior) else prior.contiguous()), pp.bins)
        if model_idx == 0:
            # train_dino.py:91-97, 128-133 without its rows: per-point slot tables + coordinate columns, summed by the first
            # ResLayer's kernel; every layer is a kernel of the library
            pred_cls, second = dino_model.heads_from_tuples(pts, desc, idx, pp.pt_off, pp.tup_off, lazy_scale=not keep, decode=draw)
        else:
            # eval.py:210-216, then train_shot.py:75-83 + :100-111; the tuple rows are gathered inside the first ResLayer's kernel
            shot_feat, normal = shot.compute_device(pts, pp.pt_off, cfg.res * 10, cfg.res * 10)
            shot_feat = ops.nan_to_zero_(shot_feat)
            normal = ops.nan_to_zero_(normal)
            extra["shot_feat"], extra["normal"] = shot_feat, normal
            feat_shot = shot_model.encode_points(shot_feat)
            pred_cls, second = shot_model.heads_from_tuples(pts, idx, feat_shot, normal, pp.pt_off, pp.tup_off,
                                                            lazy_scale=not keep, decode=draw)
        raw_cls = pred_cls
        if prior is not None and pred_cls is not None:
            pred_cls = pred_cls + prior_arr()

        def scales():
            s_ = second if keep else model.scale_head_rows(second, pp.kept_rows32(),
                                                           scatter=(pp.kept_count, pp.max_kept, scales_buf))
            return (scale_prior + 1e-3 * s_).contiguous() if scale_prior is not None else s_.contiguous()
        pred_scales = scales() if keep else scales
        pp.vote(pts, idx, None if pred_cls is None else pred_cls.contiguous(), u, pred_scales)
        if opt:
            pp.refine(pts, idx, up_sym)                                                    # eval.py:319-355
        if two and model_idx == 0:
            dino_scored.record()                     # the DINO pass' records (scale) are final
        if two and model_idx == 1:
            torch.cuda.current_stream(dev).wait_event(dino_scored)
        pp.alignment_loss(pts, idx, up_sym)                    # eval.py:358-363; the DINO pass' scale scores both passes
        if keep:
            kept.append(dict(bins=pp.bins.cpu().numpy(), mask=pp.mask.cpu().numpy().astype(bool),
                             pred_cls=pred_cls.cpu().numpy(), raw_cls=raw_cls.cpu().numpy(), pred_scales=pred_scales.cpu().numpy(), u=u.cpu().numpy(),
                             counts=pp.counts.cpu().numpy()))

    # (two streams: the persistent MLP launches leave one CU per shader engine to the other pass' kernels, cppf_mlp_reserve_cus)
    prev_reserved = ops.mlp_reserve_cus(ops.batch_mode_reserved_cus(dev) if two else 0)
    try:
        for model_idx in (0, 1):                                                           # eval.py:219
            with torch.cuda.stream(streams[model_idx]):
                one_pass(model_idx)
    finally:
        ops.mlp_reserve_cus(prev_reserved)        # an enclosing BatchMode / mlp_cus_reserved block keeps its reservation
    for st_ in streams:
        main.wait_stream(st_)
    # ---- ensemble selection (eval.py:217,365-372): strict '<' against inf, model 0 first -- on the device ---------
    pipe.select(geo_branch, visual_branch)
    records = [pipe.results_to_numpy(pipe.result_slots[m]) for m in (0, 1)]                # the 160-byte records: the first read
    for rec in records:
        bad = np.nonzero(rec["flags"] & 6)[0]
        if bad.size:
            raise RuntimeError("instances %s were not voted (flags %s: grid above cells_cap / int32)" %
                               (bad.tolist(), rec["flags"][bad].tolist()))
    chosen = pipe.results_to_numpy(pipe.selected)
    losses = pipe.losses.cpu().numpy()                                                     # [2,B] float64
    pick = chosen["pad_"][:, 0].astype(np.int64)
    best = pipe.best.cpu().numpy()
    scale = records[0]["scale"].copy()                                                     # eval.py:308-310: float32 [B,3]
    scale_norm = np.array([np.linalg.norm(s_) for s_ in scale], dtype=np.float32)          # np.linalg.norm per instance
    out = dict(records=records, selected=chosen, losses=losses, pick=pick, best=best, scale=scale.astype(np.float64),
               scale_norm=scale_norm.astype(np.float64), idx=idx, pipe=pipe, pts=pts)
    if keep:
        out["kept"] = kept
        out["shot_feat"], out["normal"] = extra["shot_feat"].cpu().numpy(), extra["normal"].cpu().numpy()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The REAL275 instance loop of the reference (eval.py:103-201, 364-412): detection results -> per-instance clouds -> poses ->
# one result record per image -> mAP.
# ---------------------------------------------------------------------------------------------------------------------
REAL_INTRINSICS = [[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]]           # eval.py:82


def load_result_list(log_dir):
    """eval.py:103-127: every results_*.pkl under log_dir (SAR-Net / Mask-RCNN detections: image_path, pred_bboxes,
    pred_masks [H,W,n], pred_class_ids, pred_scores, gt_*), a dict or a list of dicts each, flattened in file order."""
    import glob
    import pickle
    paths = sorted(glob.glob(os.path.join(log_dir, "results_*.pkl")))
    assert len(paths), "no results_*.pkl under %r" % log_dir
    final_results = []
    for path in paths:
        with open(path, "rb") as f:
            result = pickle.load(f)
        items = result if isinstance(result, list) else [result]
        assert all(isinstance(r, dict) for r in items)
        for r in items:
            if "gt_handle_visibility" not in r:
                r["gt_handle_visibility"] = np.ones_like(r["gt_class_ids"])
            else:
                assert len(r["gt_handle_visibility"]) == len(r["gt_class_ids"])
        final_results += items
    return final_results


def crop_transform(bbox, padding=0.0, out_size=256):
    """The 3x3 crop-frame -> image-frame transform of resize_crop (dataset.py:322-337) for a PIL bbox (left, upper, right,
    lower): key points of the crop are `inv(transform) @ (x, y, 1)` (eval.py:203)."""
    width, height = bbox[2] - bbox[0], bbox[3] - bbox[1]
    size = max(height, width) * (1 + padding)
    cx, cy = (bbox[2] + bbox[0]) / 2, (bbox[3] + bbox[1]) / 2
    return (np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.]])
            @ np.array([[size / out_size, 0, 0], [0, size / out_size, 0], [0, 0, 1]])
            @ np.array([[1, 0, -out_size / 2], [0, 1, -out_size / 2], [0, 0, 1.]]))


def _read_depth(path):
    from PIL import Image
    return np.array(Image.open(path)).astype(np.float64)             # cv2.imread(path, -1) of a 16-bit PNG (eval.py:139)


def image_instances(res, data_root, cfgs, seed, image_index, intrinsics=REAL_INTRINSICS, token_maps=None):
    """eval.py:133-203 for one image: yields one dict per detection that reaches the voting path -- instance index i, category,
    cloud pc float32 [n,3] (back-projected through the mask, flipped, voxel down-sampled at cfg.res, capped at 50 000 points)
    and desc float32 [n,1024] or None.  Detections of other classes and clouds wider than 1000 cells are skipped like the
    reference does (their pred_RTs stay the identity)."""
    from PIL import Image
    image_path = res["image_path"].replace("data/real/test", data_root)                    # eval.py:133
    depth = _read_depth(image_path + "_depth.png")
    masks = np.asarray(res["pred_masks"])
    rgb = None
    if os.path.exists(image_path + "_color.png"):
        rgb = np.array(Image.open(image_path + "_color.png").convert("RGB"))
    K = np.asarray(intrinsics, dtype=np.float64).reshape(3, 3)
    for i in range(len(res["pred_bboxes"])):
        cls_id = int(res["pred_class_ids"][i])
        cat = id2category.get(cls_id)
        if cat not in cfgs:                                                                # eval.py:163-165 (whitelist)
            continue
        cfg = cfgs[cat]
        mask = masks[:, :, i] != 0
        pc, (rr, cc) = ops.backproject(depth / 1000., K, mask, return_device=True)         # eval.py:185-189
        if pc.shape[0] == 0:
            continue
        inst_seed = (seed * 1000003 + image_index * 131 + i) & 0x7FFFFFFF
        keep = ops.downsample(pc, cfg.res, inst_seed, return_device=True)                   # eval.py:191-193
        pc, rr, cc = pc[keep], rr[keep.long()], cc[keep.long()]
        pc = pc.cpu().numpy()
        idxs = np.stack([rr.cpu().numpy(), cc.cpu().numpy()], -1).astype(np.int64)         # K x 2 (row, col)
        if pc.shape[0] > 50000:                                                            # eval.py:194-197
            sub = np.random.RandomState(inst_seed).randint(pc.shape[0], size=(50000,))
            pc, idxs = pc[sub], idxs[sub]
        if ((pc.max(0) - pc.min(0)).max() / cfg.res) > 1000:                               # eval.py:199-200
            continue
        desc = None
        tok = None if token_maps is None else token_maps.get("%d_%d" % (image_index, i))
        if tok is not None:
            # eval.py:177-183,202-205: crop frame of the masked RGB (its non-zero bounding box; the mask's when the colour
            # image is absent), key points = pixel (col, row) mapped into the 256 x 256 crop, descriptors = the ViT patch
            # tokens of the crop (an INPUT of the path: DINOv2 weights are not part of it) sampled there, stride 4 (dataset.py:63)
            if rgb is not None:
                masked = np.zeros_like(rgb)
                masked[mask] = rgb[mask]
                bbox = Image.fromarray(masked).getbbox()
            else:
                bbox = Image.fromarray(mask.astype(np.uint8) * 255).getbbox()
            transform = crop_transform(bbox, padding=0, out_size=256)
            kp = np.flip(idxs, -1).astype(np.float64)
            kp_local = (np.linalg.inv(transform) @ np.concatenate([kp, np.ones((kp.shape[0], 1))], -1).T).T[:, :2]
            tok = np.asarray(tok, dtype=np.float32)
            desc = ops.interpolate_features(torch.from_numpy(tok)[None], kp_local.astype(np.float32)[None], strides=4)[0].T
            desc = desc.contiguous()               # stays on the device until its batch is evaluated (run_ensemble)
        yield dict(i=i, cat=cat, pc=pc, desc=desc, pixels=idxs)


def main_nocs(setups, log_dir, data_root="NOCS/real_test", out_dir=None, desc_npz=None, angle_tol=1., imp_wt_margin=0.01,
              backproj_ratio=.1, num_pairs=50000, num_rots=180, opt=True, geo_branch=True, visual_branch=True, seed=0,
              batch_instances=16, intrinsics=None, max_images=None, debug=False, out=None):
    """eval.py:103-412 on a directory in the reference's layout: `log_dir`/results_*.pkl (detections + ground truth per image)
    and `data_root`/<scene>/<frame>_{depth,color}.png.  Instances are collected image by image exactly as the reference
    filters them, evaluated in batches per category on the GPU (run_ensemble), and written back into their image's record
    (pred_RTs, pred_scales: eval.py:143-147, 370-372); every record is pickled under out_dir with the reference's file name
    (eval.py:134, 399) and the list is scored with degree_cm_mAP (eval.py:400-412).
    desc_npz: optional .npz of DINOv2 patch-token maps float32 [1024, 64, 64] keyed "<image index>_<instance index>" (the
    crop of eval.py:177-183 at stride 4); without it seeded unit vectors stand in (the DINO branch then votes on noise)."""
    import pickle
    from cppf2_amd import metrics
    dev = ops._dev()
    final_results = load_result_list(log_dir)
    if max_images:
        final_results = final_results[:int(max_images)]
    token_maps = np.load(desc_npz) if desc_npz else None
    cfgs = {c: s[0] for c, s in setups.items()}
    K = REAL_INTRINSICS if intrinsics is None else intrinsics
    # Instances are evaluated as they come: every category keeps at most `batch_instances` pending instances (point cloud +
    # descriptors, the latter on the device) and is flushed through run_ensemble when the batch is full -- the same batches, in
    # the same order per category, as collecting the whole list first, with memory bounded by the batch (a full REAL275 run has
    # ~12 000 instances x up to 200 MB of descriptors).
    pending = {c: [] for c in setups}               # category -> [(image index, instance index, global instance id, pc, desc)]
    seen = {c: 0 for c in setups}
    evaluated = 0
    picks = {"dino": 0, "shot": 0, "none": 0}

    def flush(cat):
        nonlocal evaluated
        chunk, pending[cat] = pending[cat], []
        if not chunk:
            return
        cfg, dino_model, shot_model = setups[cat]
        descs = []
        for (_, _, g_, pc, desc) in chunk:
            if desc is None:
                # no token maps: seeded unit vectors from a CPU generator, one stream per instance -- the same numbers on every
                # device and in every round (round 4 drew them with a device generator, whose stream is not the CPU's: seeded
                # `--data=nocs` stand-in runs were not comparable with earlier ones)
                gen = torch.Generator(device="cpu").manual_seed(seed * 7919 + g_ + 1)
                desc = torch.nn.functional.normalize(torch.randn((pc.shape[0], 1024), generator=gen), dim=-1).to(dev)
            descs.append(desc)
        r = run_ensemble(cfg, dino_model, shot_model, [c_[3] for c_ in chunk], descs, seed, [c_[2] for c_ in chunk],
                         num_pairs, num_rots, angle_tol, imp_wt_margin, backproj_ratio, bool(opt), geo_branch,
                         visual_branch, cat in UP_SYM)
        for b, (n_img, i, _, _, _) in enumerate(chunk):
            evaluated += 1
            if r["pick"][b] < 0:
                picks["none"] += 1
                continue
            picks[["dino", "shot"][r["pick"][b]]] += 1
            rec = r["records"][r["pick"][b]][b]
            res = final_results[n_img]
            res["pred_RTs"][i][:3, :3] = rec["R"] * r["scale_norm"][b]                 # eval.py:370
            res["pred_RTs"][i][:3, -1] = rec["t"]                                      # eval.py:371
            if r["scale_norm"][b] > 0:
                res["pred_scales"][i] = r["scale"][b] / r["scale_norm"][b]              # eval.py:372

    gid = 0
    for n_img, res in enumerate(final_results):
        nb = len(res["pred_bboxes"])
        res["pred_RTs"] = np.stack([np.eye(4) for _ in range(nb)]) if nb else np.zeros((0, 4, 4))      # eval.py:143
        res["pred_scales"] = np.stack([np.ones((3,)) for _ in range(nb)]) if nb else np.zeros((0, 3))  # eval.py:144
        for inst in image_instances(res, data_root, cfgs, seed, n_img, K, token_maps):
            cat = inst["cat"]
            pending[cat].append((n_img, inst["i"], gid + inst["i"], inst["pc"], inst["desc"]))
            seen[cat] += 1
            if len(pending[cat]) >= int(batch_instances):
                flush(cat)
        gid += nb
    for cat in setups:
        flush(cat)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        for res in final_results:
            image_path = res["image_path"].replace("data/real/test", data_root)
            with open(os.path.join(out_dir, "_".join(image_path.split("/")[1:]) + ".pkl"), "wb") as f:     # eval.py:134,399
                pickle.dump(res, f)
    total = sum(len(r_["pred_bboxes"]) for r_ in final_results)
    iou_aps, aps = metrics.degree_cm_mAP(final_results, metrics.SYNSET_NAMES, (5, 10, 15), (5, 10, 15),
                                         np.linspace(0, 1, 101), 0.1, True)                # eval.py:400-411
    cats = [c for c in setups if seen[c]]

    def mean_over(fn):
        v = [fn(category2id[c]) for c in cats]
        v = [x for x in v if np.isfinite(x)]
        return float(np.mean(v)) if v else None
    report = dict(data="nocs", images=len(final_results), detections=total, evaluated=evaluated, skipped=total - evaluated,
                  picked=picks, categories=cats, descriptors="token maps from %s" % desc_npz if desc_npz else "seeded unit vectors (no DINOv2 tokens given)",
                  pose_AP={"%ddeg_%dcm" % (d_, s_): mean_over(lambda c, i_=i_, j_=j_: aps[c, i_, j_])
                           for i_, d_ in enumerate((5, 10, 15)) for j_, s_ in enumerate((5, 10, 15))},
                  iou_AP={"IoU%d" % t_: mean_over(lambda c, t_=t_: iou_aps[c, t_]) for t_ in (25, 50, 75)})
    print(json.dumps(report))
    if out:
        with open(out, "w") as f:
            json.dump(report, f)
    report["final_results"] = final_results
    return report



def _teacher_prior(canon, dev):
    canon = torch.from_numpy(canon).to(dev)
    kb = torch.arange(32, device=dev, dtype=torch.float32)

    def prior(idx, base):
        coords = canon[(idx[:, :2].long() + base[:, None]).reshape(-1)].reshape(-1, 6)
        pos = (coords.clamp(-0.5, 0.5) + 0.5) * 31.0
        return ops.BinPrior(pos.contiguous(), 1.0 / 0.6)          # generated inside the fused bin draw; .dense() where an array is needed
    return prior


def main(angle_tol=1., imp_wt_margin=0.01, backproj_ratio=.1, num_pairs=50000, num_rots=180, opt=True, debug=False,
         use_grounded_sam=False, geo_branch=True, visual_branch=True, data="synthetic", num_scenes=8, num_points=4096,
         category=None, categories=None, seed=0, ckpt_dir=None, ckpt_shot=None, ckpt_dino=None, depth=None, mask=None,
         intrinsics=None, depth_scale=1000.0, out=None, out_pkl=None, log_dir=None, data_root="NOCS/real_test", out_dir=None,
         desc_npz=None, batch_instances=16, max_images=None):
    custom = False
    if categories is None:
        if category:
            categories = [category]
        elif data == "depth":
            # a single depth + mask pair is one instance; without --category it is an instance-level object like the
            # reference's example (a YCB object: config/custom.yaml, no category group, full rotation)
            categories, custom = ["custom"], True
        else:
            categories = [id2category[i] for i in range(1, 7)]                             # eval.py:87-90
    elif isinstance(categories, str):
        categories = [c for c in categories.replace(" ", "").split(",") if c]
    categories = [c for c in categories if c in WHITELIST or custom]
    dev = ops._dev()
    torch.manual_seed(seed)
    # eval.py:84-101: models and cfgs of every category up front
    if custom:
        setups = {"custom": load_custom(ckpt_shot, ckpt_dino, device=dev)}
    else:
        setups = {c: load_category(c, ckpt_dir, ckpt_shot, ckpt_dino, device=dev) for c in categories}
    if data == "nocs":
        assert log_dir, "--data=nocs needs --log_dir (the directory of results_*.pkl, eval.py:72-76)"
        return main_nocs(setups, log_dir, data_root, out_dir, desc_npz, angle_tol, imp_wt_margin, backproj_ratio, num_pairs,
                         num_rots, opt, geo_branch, visual_branch, seed, batch_instances, intrinsics, max_images, debug, out)

    from cppf2_amd import metrics
    summary, all_cls, all_RT, all_scale, all_gt, all_gt_scale = [], [], [], [], [], []
    inst = 0
    for ci, cat in enumerate(categories):
        cfg, dino_model, shot_model = setups[cat]
        up_sym = cat in UP_SYM or bool(cfg.get("up_sym", False))
        # ---- instances ---------------------------------------------------------------------------
        if data == "depth":
            from PIL import Image
            d = np.array(Image.open(depth)).astype(np.float64) / float(depth_scale)
            m = np.array(Image.open(mask))
            m = (m[..., 0] if m.ndim == 3 else m) > 0
            K = np.array(intrinsics if intrinsics is not None else
                         [[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float64).reshape(3, 3)
            pc, _ = ops.backproject(d, K, m, return_device=True)               # eval.py:185-189 (flip + f32 cast folded in)
            pc = pc[ops.downsample(pc, cfg.res, seed, return_device=True)].cpu().numpy()   # eval.py:192
            if pc.shape[0] > 50000:
                pc = pc[np.random.RandomState(seed).randint(pc.shape[0], size=50000)]
            scenes = [dict(pc=pc, pc_canon=None, R=None, t=None)]
        else:
            scenes = [synth.make_scene(seed, inst + s, num_points) for s in range(num_scenes)]
        made = len(scenes)
        keep_ids = [inst + j for j, s in enumerate(scenes) if ((s["pc"].max(0) - s["pc"].min(0)).max() / cfg.res) <= 1000]
        scenes = [s for s in scenes if ((s["pc"].max(0) - s["pc"].min(0)).max() / cfg.res) <= 1000]     # eval.py:200
        B = len(scenes)
        if B == 0:
            inst += made
            continue
        scene_ids = keep_ids          # a dropped instance does not shift the others' seeds (tuple / uniform streams = scene seed)
        # DINOv2 features are inputs to the path (weights absent): seeded unit vectors stand in for them
        g = torch.Generator(device="cpu").manual_seed(seed + 1 + ci)
        descs = [torch.nn.functional.normalize(torch.randn((s["pc"].shape[0], 1024), generator=g), dim=-1).numpy()
                 for s in scenes]
        priors = scale_priors = None
        if scenes[0]["pc_canon"] is not None:
            priors = _teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
            scale_priors = np.stack([s["extent"] for s in scenes])
        r = run_ensemble(cfg, dino_model, shot_model, [s["pc"] for s in scenes], descs, seed, scene_ids, num_pairs,
                         num_rots, angle_tol, imp_wt_margin, backproj_ratio, bool(opt), geo_branch, visual_branch,
                         up_sym, priors, scale_priors=scale_priors)
        cls_id = category2id.get(cat, 0)
        for b in range(B):
            RT, sc = np.eye(4), np.ones(3)                                      # eval.py:143-144 defaults
            item = dict(scene=scene_ids[b], category=cat, model=None)
            if r["pick"][b] >= 0:                                               # eval.py:367-372
                rec = r["records"][r["pick"][b]][b]
                RT[:3, :3] = rec["R"] * r["scale_norm"][b]
                RT[:3, 3] = rec["t"]
                if r["scale_norm"][b] > 0:
                    sc = r["scale"][b] / r["scale_norm"][b]
                item.update(model=["dino", "shot"][r["pick"][b]], loss=float(r["best"][b]),
                            losses=[float(r["losses"][0][b]), float(r["losses"][1][b])], pred_RT=RT.tolist(),
                            pred_scale=sc.tolist())
                if scenes[b]["R"] is not None:
                    item["tr_err_cm"] = float(np.linalg.norm(rec["t"] - scenes[b]["t"]) * 100)
                    item["rot_err_deg"] = geometry.rot_err_deg(rec["R"], scenes[b]["R"], up_sym)
            summary.append(item)
            all_cls.append(cls_id); all_RT.append(RT); all_scale.append(sc)
            if scenes[b]["R"] is not None:
                # synthetic instances carry their pose and box: NOCS convention, rotation scaled by the box diagonal and
                # the extents normalised by it (what eval.py:370-372 builds from the prediction)
                gt = np.eye(4)
                gt[:3, :3], gt[:3, 3] = scenes[b]["R"] * scenes[b]["diag"], scenes[b]["t"]
                all_gt.append(gt)
                all_gt_scale.append(scenes[b]["extent"] / scenes[b]["diag"])
        inst += made

    report = dict(categories=categories, instances=len(summary),
                  opt_refinement="100 Adam steps (cppf_refine_pose)" if opt else "off", results=summary)
    if len(categories) == 1:
        report["category"] = categories[0]
    scored = [s for s in summary if "rot_err_deg" in s]
    if scored:
        report["acc_5deg_5cm"] = float(np.mean([s["rot_err_deg"] < 5 and s["tr_err_cm"] < 5 for s in scored]))
        report["acc_5deg_5cm_per_category"] = {
            c: float(np.mean([s["rot_err_deg"] < 5 and s["tr_err_cm"] < 5 for s in scored if s["category"] == c]))
            for c in categories if any(s["category"] == c for s in scored)}
    # the reference's per-image result record (eval.py:143-147, 370-372, 399): pred_RTs [n,4,4] (rotation scaled by the
    # scale norm), pred_scales [n,3] (normalised); all instances of the run form one record, like those of one image
    n = len(summary)
    gt = {}
    if n and len(all_gt) == n:
        gt = dict(gt_class_ids=np.array(all_cls), gt_RTs=np.stack(all_gt), gt_scales=np.stack(all_gt_scale))
    record = metrics.make_result_record(np.array(all_cls, dtype=np.int64), np.stack(all_RT) if n else np.zeros((0, 4, 4)),
                                        np.stack(all_scale) if n else np.zeros((0, 3)), None, **gt)
    if gt:
        # eval.py:400-411: degree / cm AP over the instances matched at 3-D IoU > 0.1, and the 3-D IoU AP itself
        thr = np.linspace(0, 1, 101)
        iou_aps, aps = metrics.degree_cm_mAP([record], metrics.SYNSET_NAMES, (5, 10, 15), (5, 10, 15), thr, 0.1, True)
        scored_cats = sorted({s_["category"] for s_ in summary if s_["category"] in category2id})

        def _mean(vals):                      # categories without a scored instance have NaN APs: they do not enter the mean
            vals = [v for v in vals if np.isfinite(v)]
            return float(np.mean(vals)) if vals else None
        report["pose_AP"] = {"%ddeg_%dcm" % (d_, s_): _mean([aps[category2id[c], i_, j_] for c in scored_cats])
                             for i_, d_ in enumerate((5, 10, 15)) for j_, s_ in enumerate((5, 10, 15))}
        report["pose_AP_per_category"] = {c: {"%ddeg_%dcm" % (d_, s_): float(aps[category2id[c], i_, j_])
                                              for i_, d_ in enumerate((5, 10, 15)) for j_, s_ in enumerate((5, 10, 15))}
                                          for c in scored_cats}
        report["iou_AP"] = {"IoU%d" % t_: _mean([iou_aps[category2id[c], t_] for c in scored_cats]) for t_ in (25, 50, 75)}
    if out_pkl:
        import pickle
        with open(out_pkl, "wb") as f:
            pickle.dump(record, f)
    print(json.dumps(report if debug else {k_: v for k_, v in report.items() if k_ != "results"}))
    if out:
        with open(out, "w") as f:
            json.dump(report, f)
    return report


if __name__ == "__main__":
    kwargs = {}
    for a in sys.argv[1:]:
        if a.startswith("--") and "=" in a:
            k_, v_ = a[2:].split("=", 1)
            kwargs[k_] = _flag(v_)
        elif a.startswith("--"):
            kwargs[a[2:]] = True
    main(**kwargs)
