"""Scoring of the path's output, host-side NumPy like the reference's:
  * the 5 deg / 5 cm pose criterion: rotation/translation error between two similarity transforms with
    the category symmetries of the NOCS toolkit (utils/util.py:588-663);
  * the per-image result record eval.py writes (eval.py:143-147, 399) and the scores computed from a list of them
    (utils/util.py:2610-2955): the degree / cm pose AP (pose_mAP; degree_cm_mAP with
    use_matches_for_pose as eval.py:400-411 calls it) and the 3-D IoU AP over oriented boxes (box_iou_3d, iou_3d:
    utils/util.py:475-547 with utils/iou.py:15-216 and utils/box.py:41-280).  All pinned by the reference toolkit
    itself (tests/golden/make_golden_map.py -> map_results.pkl)."""
from __future__ import annotations

import numpy as np

SYNSET_NAMES = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]       # eval.py:400-407
_AXIS_SYMMETRIC = ("bottle", "can", "bowl")
_AXIS_SYMMETRIC_IF_NO_HANDLE = ("mug", "chair", "bathtub", "bookshelf", "bed", "sofa", "table")
_HALF_TURN_SYMMETRIC = ("phone", "eggbox", "glue")


def _unit_rotation(RT):
    R = RT[:3, :3]
    return R / np.cbrt(np.linalg.det(R))


def rt_degree_cm(RT_1, RT_2, class_name, handle_visibility=1, clip=False):
    """Returns (theta in degrees, shift in centimetres), or None if either pose is missing.
    clip=False reproduces the toolkit exactly (arccos of a cosine that rounding pushed above 1 is NaN, e.g. for
    identical rotations); clip=True clamps the cosine to [-1, 1] first."""
    if RT_1 is None or RT_2 is None:
        return None
    RT_1, RT_2 = np.asarray(RT_1, dtype=np.float64), np.asarray(RT_2, dtype=np.float64)
    last = np.array([0.0, 0.0, 0.0, 1.0])
    if not (np.array_equal(RT_1[3], last) and np.array_equal(RT_2[3], last)):
        raise ValueError("last row of a pose must be [0, 0, 0, 1]")
    R1, R2 = _unit_rotation(RT_1), _unit_rotation(RT_2)
    acos = (lambda c: np.arccos(np.clip(c, -1.0, 1.0))) if clip else np.arccos
    about_y = class_name in _AXIS_SYMMETRIC or (class_name in _AXIS_SYMMETRIC_IF_NO_HANDLE and handle_visibility == 0)
    if about_y:
        y1, y2 = R1[:, 1], R2[:, 1]                       # R @ [0,1,0]
        theta = acos(y1.dot(y2) / (np.linalg.norm(y1) * np.linalg.norm(y2)))
    elif class_name in _HALF_TURN_SYMMETRIC:
        flip = np.diag([-1.0, 1.0, -1.0])
        theta = min(acos((np.trace(R1 @ R2.T) - 1) / 2), acos((np.trace(R1 @ flip @ R2.T) - 1) / 2))
    else:
        theta = acos((np.trace(R1 @ R2.T) - 1) / 2)
    return float(theta * 180 / np.pi), float(np.linalg.norm(RT_1[:3, 3] - RT_2[:3, 3]) * 100)


def match_rate(RTs_a, RTs_b, class_name, deg=5.0, cm=5.0):
    """Fraction of pose pairs within (deg, cm) of each other."""
    ok = []
    for a, b in zip(RTs_a, RTs_b):
        r = rt_degree_cm(a, b, class_name, clip=True)
        ok.append(r is not None and r[0] <= deg and r[1] <= cm)
    return float(np.mean(ok)) if ok else float("nan")


# ----------------------------------------------------------------------------------------------
# result records and the degree / cm pose mAP
# ----------------------------------------------------------------------------------------------
RESULT_KEYS = ("gt_class_ids", "gt_RTs", "gt_scales", "gt_handle_visibility",
               "pred_class_ids", "pred_RTs", "pred_scales", "pred_scores")


def make_result_record(pred_class_ids, pred_RTs, pred_scales, pred_scores=None, gt_class_ids=(), gt_RTs=None,
                       gt_scales=None, gt_handle_visibility=None, **extra):
    """One image's record in the layout the reference pickles (eval.py:143-147: `pred_RTs` float64[n,4,4] with the
    scale norm folded into the rotation, `pred_scales` float64[n,3] normalised; eval.py:399 dumps the dict)."""
    n, m = len(pred_class_ids), len(gt_class_ids)
    rec = dict(pred_class_ids=np.asarray(pred_class_ids, dtype=np.int32),
               pred_RTs=np.asarray(pred_RTs, dtype=np.float64).reshape(n, 4, 4),
               pred_scales=np.asarray(pred_scales, dtype=np.float64).reshape(n, 3),
               pred_scores=np.ones(n) if pred_scores is None else np.asarray(pred_scores, dtype=np.float64),
               gt_class_ids=np.asarray(gt_class_ids, dtype=np.int32),
               gt_RTs=np.zeros((0, 4, 4)) if gt_RTs is None else np.asarray(gt_RTs, dtype=np.float64).reshape(m, 4, 4),
               gt_scales=np.zeros((0, 3)) if gt_scales is None else np.asarray(gt_scales, dtype=np.float64).reshape(m, 3),
               gt_handle_visibility=np.ones(m, np.int32) if gt_handle_visibility is None
               else np.asarray(gt_handle_visibility, dtype=np.int32))
    rec.update(extra)
    return rec


def _unit_scale(RTs):
    """RTs with the isotropic scale divided out of the rotation block (utils/util.py:2619-2620, 2632-2633)."""
    RTs = np.array(RTs, dtype=np.float64).reshape(-1, 4, 4)
    for RT in RTs:
        RT[:3, :3] = RT[:3, :3] / (np.cbrt(np.linalg.det(RT[:3, :3])) + 1e-7)
    return RTs


def average_precision(pred_match, pred_scores, gt_match):
    """VOC-style AP from per-prediction matches (utils/util.py:1757-1782): predictions by descending score, precision
    envelope, area over the recall steps.  NaN when there is no ground truth and nothing matches (0/0), like the toolkit."""
    pred_match, pred_scores = np.asarray(pred_match), np.asarray(pred_scores)
    assert pred_match.shape[0] == pred_scores.shape[0]
    hit = (pred_match[np.argsort(pred_scores)[::-1]] > -1)
    tp = np.cumsum(hit)
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.concatenate([[0.0], tp / (np.arange(len(hit)) + 1), [0.0]])
        recall = np.concatenate([[0.0], tp.astype(np.float32) / len(gt_match), [1.0]])
    precision = np.maximum.accumulate(precision[::-1])[::-1]
    step = np.where(recall[:-1] != recall[1:])[0] + 1
    return np.sum((recall[step] - recall[step - 1]) * precision[step])


def pose_overlaps(gt_class_ids, gt_RTs, gt_handle_visibility, pred_RTs, synset_names):
    """[num_pred, num_gt, 2] (degrees, centimetres) with the ground truth's class symmetry (utils/util.py:1785-1808)."""
    out = np.zeros((len(pred_RTs), len(gt_class_ids), 2))
    for i, p in enumerate(pred_RTs):
        for j, g in enumerate(gt_RTs):
            out[i, j] = rt_degree_cm(p, g, synset_names[int(gt_class_ids[j])], gt_handle_visibility[j])
    return out


def match_by_degree_cm(overlaps, pred_class_ids, gt_class_ids, degree_thresholds, shift_thresholds):
    """Greedy matching per (degree, shift) threshold pair (utils/util.py:1883-1926): predictions in the given order,
    candidates by ascending degrees + centimetres, a ground truth is used once.  Returns (gt_matches, pred_matches),
    -1 = unmatched."""
    nd, ns, npred, ngt = len(degree_thresholds), len(shift_thresholds), len(pred_class_ids), len(gt_class_ids)
    pred_m, gt_m = -np.ones((nd, ns, npred)), -np.ones((nd, ns, ngt))
    if npred == 0 or ngt == 0:
        return gt_m, pred_m
    order = np.argsort(overlaps.sum(-1), axis=1)
    same = np.asarray(pred_class_ids)[:, None] == np.asarray(gt_class_ids)[None, :]
    for d, dt in enumerate(degree_thresholds):
        for s, st in enumerate(shift_thresholds):
            ok = same & ~(overlaps[..., 0] > dt) & ~(overlaps[..., 1] > st)      # NaN errors compare False, as there
            for i in range(npred):
                for j in order[i]:
                    if gt_m[d, s, j] < 0 and ok[i, j]:
                        gt_m[d, s, j], pred_m[d, s, i] = i, j
                        break
    return gt_m, pred_m


def pose_mAP(final_results, synset_names=SYNSET_NAMES, degree_thresholds=(5, 10, 15), shift_thresholds=(5, 10, 15)):
    """Degree / cm pose AP per class and their mean (compute_degree_cm_mAP's `pose_aps`, utils/util.py:2736-2955, with
    use_matches_for_pose=False as in its signature).  Returns float64[num_classes + 1, len(deg) + 1, len(shift) + 1]:
    the trailing threshold of each axis is the toolkit's catch-all (360 deg, 100 cm), row 0 is the background class
    (zeros), the last row the mean over classes."""
    degs, shifts = list(degree_thresholds) + [360], list(shift_thresholds) + [100]
    ncls = len(synset_names)
    pm = [[] for _ in range(ncls)]
    ps = [[] for _ in range(ncls)]
    gm = [[] for _ in range(ncls)]
    for res in final_results:
        gt_cls = np.asarray(res["gt_class_ids"]).astype(np.int32)
        gt_RTs = _unit_scale(res["gt_RTs"])
        gt_vis = np.asarray(res["gt_handle_visibility"])
        pr_cls = np.asarray(res["pred_class_ids"])
        pr_RTs = _unit_scale(res["pred_RTs"])
        pr_sc = np.asarray(res["pred_scores"], dtype=np.float64)
        if len(gt_cls) == 0 and len(pr_cls) == 0:
            continue
        for c in range(1, ncls):
            g = gt_cls == c
            p = pr_cls == c
            c_gt_RTs = gt_RTs[g]
            c_vis = gt_vis[g] if synset_names[c] == "mug" else np.ones(int(g.sum()))
            c_scores, c_RTs = pr_sc[p], pr_RTs[p]
            by_score = np.argsort(c_scores)[::-1]                      # the order compute_3d_matches leaves behind
            c_scores, c_RTs = c_scores[by_score], c_RTs[by_score]
            ov = pose_overlaps(gt_cls[g], c_gt_RTs, c_vis, c_RTs, synset_names)
            g_match, p_match = match_by_degree_cm(ov, np.full(len(c_RTs), c), gt_cls[g], degs, shifts)
            pm[c].append(p_match)
            ps[c].append(np.tile(c_scores, (len(degs), len(shifts), 1)))
            gm[c].append(g_match)
    aps = np.zeros((ncls + 1, len(degs), len(shifts)))
    for c in range(1, ncls):
        if not pm[c]:
            aps[c] = np.nan
            continue
        P, S, G = np.concatenate(pm[c], -1), np.concatenate(ps[c], -1), np.concatenate(gm[c], -1)
        for i in range(len(degs)):
            for j in range(len(shifts)):
                aps[c, i, j] = average_precision(P[i, j], S[i, j], G[i, j])
    aps[-1] = np.mean(aps[1:-1], axis=0)
    return aps


# ----------------------------------------------------------------------------------------------
# 3-D IoU of oriented boxes and its AP (utils/util.py:475-547, 1665-1754, 2610-2733)
# ----------------------------------------------------------------------------------------------
_CORNERS = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)])
_EDGES = np.array([(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1])   # 12 box edges


def _clip_edges_to_aabb(P, Q, half):
    """Segments P[i] -> Q[i] clipped to the axis-aligned box [-half, half] (slab method).  Returns the end points of the
    non-empty clipped segments: box corners of the segment's box that lie inside, and edge x face intersections."""
    d = Q - P
    t0, t1 = np.zeros(len(P)), np.ones(len(P))
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):
            near, far = (-half[a] - P[:, a]) / d[:, a], (half[a] - P[:, a]) / d[:, a]
            lo, hi = np.minimum(near, far), np.maximum(near, far)
            par = d[:, a] == 0
            inside = np.abs(P[:, a]) <= half[a]
            lo = np.where(par, np.where(inside, -np.inf, np.inf), lo)
            hi = np.where(par, np.where(inside, np.inf, -np.inf), hi)
            t0, t1 = np.maximum(t0, lo), np.minimum(t1, hi)
    keep = t0 <= t1
    return np.concatenate([P[keep] + t0[keep, None] * d[keep], P[keep] + t1[keep, None] * d[keep]])


def box_iou_3d(R1, t1, s1, R2, t2, s2):
    """Exact IoU of two oriented boxes (rotation R, centre t, edge lengths s).  The intersection of two convex
    polytopes has three kinds of vertices -- corners of one inside the other and edge x face crossings -- all of which
    fall out of clipping each box's 12 edges to the other box in that box's own frame; its volume is the convex hull's
    (same quantity as utils/iou.py:22-35, which clips face polygons instead).  Degenerate contact (no volume) -> 0."""
    from scipy.spatial import ConvexHull, QhullError
    R1, R2 = np.asarray(R1, dtype=np.float64), np.asarray(R2, dtype=np.float64)
    t1, t2 = np.asarray(t1, dtype=np.float64).reshape(3), np.asarray(t2, dtype=np.float64).reshape(3)
    s1, s2 = np.asarray(s1, dtype=np.float64).reshape(3), np.asarray(s2, dtype=np.float64).reshape(3)
    pts = []
    for (Ra, ta, sa), (Rb, tb, sb) in (((R1, t1, s1), (R2, t2, s2)), ((R2, t2, s2), (R1, t1, s1))):
        world = (_CORNERS * sa) @ Ra.T + ta                       # corners of box a
        local = (world - tb) @ Rb                                 # in box b's frame (Rb orthonormal)
        clipped = _clip_edges_to_aabb(local[_EDGES[:, 0]], local[_EDGES[:, 1]], sb / 2)
        if len(clipped):
            pts.append(clipped @ Rb.T + tb)
    if not pts:
        return 0.0
    pts = np.concatenate(pts)
    try:
        inter = ConvexHull(pts).volume
    except (QhullError, ValueError):
        return 0.0
    v1, v2 = abs(np.linalg.det(R1)) * np.prod(s1), abs(np.linalg.det(R2)) * np.prod(s2)
    return float(inter / (v1 + v2 - inter))


def _y_rotation(theta):
    c, s_ = np.cos(theta), np.sin(theta)
    return np.array([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]])


def iou_3d(RT_1, RT_2, scales_1, scales_2, handle_visibility, class_name_1, class_name_2):
    """compute_3d_iou_new (utils/util.py:475-547): the isotropic scale is divided out of both rotations; for the
    classes that are symmetric about y (bottle / bowl / can, and a mug whose handle is hidden) the first box is tried
    in 36 rotations about its y axis and the best overlap counts."""
    if RT_1 is None or RT_2 is None:
        return -1

    def unit(RT):
        RT = np.array(RT, dtype=np.float64)
        RT[:3, :3] = RT[:3, :3] / np.cbrt(np.linalg.det(RT[:3, :3]))
        return RT
    RT_2 = unit(RT_2)
    sym = (class_name_1 in _AXIS_SYMMETRIC and class_name_1 == class_name_2) or \
          (class_name_1 == "mug" and class_name_1 == class_name_2 and handle_visibility == 0)
    best = 0.0
    for i in range(36 if sym else 1):
        A = unit(np.asarray(RT_1, dtype=np.float64) @ _y_rotation(2 * np.pi * i / 36.0)) if sym else unit(RT_1)
        best = max(best, box_iou_3d(A[:3, :3], A[:3, 3], scales_1, RT_2[:3, :3], RT_2[:3, 3], scales_2))
    return best


def match_by_iou(overlaps, pred_class_ids, gt_class_ids, iou_thresholds):
    """compute_3d_matches' greedy loop (utils/util.py:1720-1753): predictions in the given (score) order, candidates by
    descending IoU, a ground truth is used once, strict '>' against the threshold.  Returns (gt_matches, pred_matches)."""
    npred, ngt = overlaps.shape
    pred_m, gt_m = -np.ones((len(iou_thresholds), npred)), -np.ones((len(iou_thresholds), ngt))
    order = [np.argsort(overlaps[i])[::-1] for i in range(npred)]
    for s, thr in enumerate(iou_thresholds):
        for i in range(npred):
            for j in order[i]:
                if gt_m[s, j] > -1:
                    continue
                iou = overlaps[i, j]
                if iou < thr:
                    break
                if pred_class_ids[i] != gt_class_ids[j]:
                    continue
                if iou > thr:
                    gt_m[s, j], pred_m[s, i] = i, j
                    break
    return gt_m, pred_m


def degree_cm_mAP(final_results, synset_names=SYNSET_NAMES, degree_thresholds=(5, 10, 15), shift_thresholds=(5, 10, 15),
                  iou_3d_thresholds=tuple(np.linspace(0, 1, 101)), iou_pose_thres=0.1, use_matches_for_pose=False):
    """compute_degree_cm_mAP (utils/util.py:2736-2955) without the plots: returns (iou_3d_aps [num_classes + 1, len(iou
    thresholds)], pose_aps [num_classes + 1, len(deg) + 1, len(shift) + 1]).  With use_matches_for_pose (what eval.py:
    400-411 passes) the pose AP is taken over the predictions / ground truths matched at 3-D IoU > iou_pose_thres."""
    degs, shifts = list(degree_thresholds) + [360], list(shift_thresholds) + [100]
    iou_thr = list(iou_3d_thresholds)
    if use_matches_for_pose:
        assert iou_pose_thres in iou_thr
    ncls = len(synset_names)
    acc = {k: [[] for _ in range(ncls)] for k in ("ipm", "ips", "igm", "ppm", "pps", "pgm")}
    for res in final_results:
        gt_cls = np.asarray(res["gt_class_ids"]).astype(np.int32)
        gt_RTs = np.array(res["gt_RTs"], dtype=np.float64).reshape(-1, 4, 4)
        gt_scales = np.array(res["gt_scales"], dtype=np.float64).reshape(-1, 3)
        gt_vis = np.asarray(res["gt_handle_visibility"])
        gnorm = np.array([np.cbrt(np.linalg.det(RT[:3, :3])) for RT in gt_RTs]).reshape(-1)          # util.py:2619-2621
        gt_RTs[:, :3, :3] = gt_RTs[:, :3, :3] / (gnorm[:, None, None] + 1e-7)
        gt_scales = gt_scales * gnorm[:, None]
        pr_cls = np.asarray(res["pred_class_ids"])
        pr_RTs = np.array(res["pred_RTs"], dtype=np.float64).reshape(-1, 4, 4)
        pr_scales = np.array(res["pred_scales"], dtype=np.float64).reshape(-1, 3)
        pr_sc = np.asarray(res["pred_scores"], dtype=np.float64)
        if len(pr_RTs):
            pnorm = np.array([np.cbrt(np.linalg.det(RT[:3, :3])) for RT in pr_RTs])                  # util.py:2632-2634
            pr_RTs[:, :3, :3] = pr_RTs[:, :3, :3] / (pnorm[:, None, None] + 1e-7)
            pr_scales = pr_scales * pnorm[:, None]
        if len(gt_cls) == 0 and len(pr_cls) == 0:
            continue
        for c in range(1, ncls):
            g, p = gt_cls == c, pr_cls == c
            c_gt_cls, c_gt_RTs, c_gt_scales = gt_cls[g], gt_RTs[g], gt_scales[g]
            c_vis = gt_vis[g] if synset_names[c] == "mug" else np.ones(int(g.sum()))
            by_score = np.argsort(pr_sc[p])[::-1]                                                     # util.py:1688
            c_scores, c_RTs, c_pscales = pr_sc[p][by_score], pr_RTs[p][by_score], pr_scales[p][by_score]
            c_pcls = np.full(len(c_RTs), c)
            ov = np.zeros((len(c_RTs), len(c_gt_cls)), dtype=np.float32)                             # util.py:1713
            for i in range(len(c_RTs)):
                for j in range(len(c_gt_cls)):
                    ov[i, j] = iou_3d(c_RTs[i], c_gt_RTs[j], c_pscales[i], c_gt_scales[j], c_vis[j], synset_names[c],
                                      synset_names[c])
            ig, ip = match_by_iou(ov, c_pcls, c_gt_cls, iou_thr)
            acc["ipm"][c].append(ip)
            acc["ips"][c].append(np.tile(c_scores, (len(iou_thr), 1)))
            acc["igm"][c].append(ig)
            if use_matches_for_pose:                                                                  # util.py:2693-2712
                k = iou_thr.index(iou_pose_thres)
                keep_p, keep_g = ip[k] > -1, ig[k] > -1
                c_scores, c_RTs, c_pcls = c_scores[keep_p], c_RTs[keep_p], c_pcls[keep_p]
                c_gt_cls, c_gt_RTs, c_vis = c_gt_cls[keep_g], c_gt_RTs[keep_g], c_vis[keep_g]
            pov = pose_overlaps(c_gt_cls, c_gt_RTs, c_vis, c_RTs, synset_names)
            pg, pp = match_by_degree_cm(pov, c_pcls, c_gt_cls, degs, shifts)
            acc["ppm"][c].append(pp)
            acc["pps"][c].append(np.tile(c_scores, (len(degs), len(shifts), 1)))
            acc["pgm"][c].append(pg)
    iou_aps = np.zeros((ncls + 1, len(iou_thr)))
    pose_aps = np.zeros((ncls + 1, len(degs), len(shifts)))
    for c in range(1, ncls):
        if not acc["ipm"][c]:
            iou_aps[c], pose_aps[c] = np.nan, np.nan
            continue
        P, S, G = (np.concatenate(acc[k][c], -1) for k in ("ipm", "ips", "igm"))
        for s_ in range(len(iou_thr)):
            iou_aps[c, s_] = average_precision(P[s_], S[s_], G[s_])
        P, S, G = (np.concatenate(acc[k][c], -1) for k in ("ppm", "pps", "pgm"))
        for i in range(len(degs)):
            for j in range(len(shifts)):
                pose_aps[c, i, j] = average_precision(P[i, j], S[i, j], G[i, j])
    iou_aps[-1] = np.mean(iou_aps[1:-1], axis=0)
    pose_aps[-1] = np.mean(pose_aps[1:-1], axis=0)
    return iou_aps, pose_aps
