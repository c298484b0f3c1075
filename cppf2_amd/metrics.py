"""Scoring of the path's output, host-side NumPy like the reference's:
  * the 5 deg / 5 cm pose criterion: rotation/translation error between two similarity transforms with
    the category symmetries of the NOCS toolkit (utils/util.py:588-663);
  * the per-image result record eval.py writes (eval.py:143-147, 399) and the degree / cm pose mAP computed from a list
    of them (utils/util.py:2610-2955 with use_matches_for_pose=False).  The 3-D IoU AP of the same
    toolkit needs an oriented-box intersection (utils/box.py, utils/iou.py) and is not built."""
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
