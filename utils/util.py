"""`utils.util` of the reference for the names its entry points import from it -- eval.py:5 / demo.py:5
(downsample, backproject, dilate_mask, fibonacci_sphere, real2prob, prob2real, calculate_2d_projections, draw, get_3d_bbox,
process_data, transform_coordinates_3d, compute_degree_cm_mAP), train_shot.py:16 / train_dino.py:18 (real2prob, prob2real),
dataset.py:4,9,20 -- so those import lines resolve unchanged with this repository first on sys.path.

The functions on or next to the voting path run through libcppf_hip.so (backproject, downsample: SURVEY.md 8f-2) or are the
host code the reference has there (fibonacci_sphere: 8a-10, the mAP scorer: 8f-4); the rest are the small NumPy helpers of the
NOCS toolkit that eval.py's visualisation branch and dataset.py call, written out here so the names are real functions, not
placeholders.  Argument order, dtypes and return conventions are the reference's (file:line in each docstring).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from cppf2_amd import metrics as _metrics
from cppf2_amd import ops as _ops

fibonacci_sphere = _ops.fibonacci_sphere          # utils/util.py:191-207: list of 3-tuples, Python float64 math


def downsample(pc, res):
    """utils/util.py:39-46: indices of one uniformly random point per `res` voxel (int64 ndarray, ascending; the reference's are
    in open3d's voxel order -- callers only index with them, eval.py:193-195).  cppf_voxel_downsample on the GPU; the draw is
    Philox(point index), not NumPy's global RandomState."""
    return _ops.downsample(pc, res)


def backproject(depth, intrinsics, instance_mask):
    """utils/util.py:2586-2607: (pts float64[n,3] with x and y negated, (rows, cols)) for the masked pixels with depth > 0, in
    np.where's row-major order.  cppf_backproject64 on the GPU: the reference's float64 operations in the reference's order, so
    the array is the reference's bit for bit (callers negate x, y back and cast to float32, eval.py:187-189)."""
    return _ops.backproject_reference(depth, intrinsics, instance_mask)


def real2prob(val, max_val, num_bins, circular=False):
    """utils/util.py:215-251: soft one-hot of `val` over `num_bins` knots (weight 1-frac on the lower knot, frac on the upper);
    circular: bin centres at (i + 0.5) * max_val / num_bins with wrap-around.  torch tensors or ndarrays, like the reference."""
    is_torch = isinstance(val, torch.Tensor)
    if circular:
        interval = max_val / num_bins
        shifted = val.clone() if is_torch else val.copy()
        shifted[val < interval / 2] += max_val
        res = real2prob(shifted - interval / 2, max_val, num_bins + 1)
        res[..., 0] += res[..., -1]
        return res[..., :-1]
    interval = max_val / (num_bins - 1)
    x = val / interval
    if is_torch:
        low = torch.clamp(torch.floor(x).long(), max=num_bins - 2)
        res = torch.zeros((*val.shape, num_bins), dtype=val.dtype, device=val.device)
        wl = 1.0 - (x - low)
        res.scatter_(-1, low[..., None], wl[..., None])
        res.scatter_(-1, (low + 1)[..., None], (1.0 - wl)[..., None])
        return res
    low = np.minimum(np.floor(x).astype(np.int64), num_bins - 2)
    res = np.zeros((*val.shape, num_bins), dtype=val.dtype)
    wl = 1.0 - (x - low)
    np.put_along_axis(res, low[..., None], wl[..., None], -1)
    np.put_along_axis(res, (low + 1)[..., None], (1.0 - wl)[..., None], -1)
    return res


def prob2real(prob, max_val, num_bins, circular=False):
    """utils/util.py:254-272: expectation of the knot positions under `prob`; circular: angle of the mean direction of the bin
    centres, remapped to [0, 2 pi)."""
    is_torch = isinstance(prob, torch.Tensor)
    knots = torch.arange(num_bins).to(prob) if is_torch else np.arange(num_bins)
    if not circular:
        return (prob * knots * max_val / (num_bins - 1)).sum(-1)
    interval = max_val / num_bins
    ang = knots * interval + interval / 2
    if is_torch:
        res = torch.atan2((prob * torch.sin(ang)).sum(-1), (prob * torch.cos(ang)).sum(-1))
    else:
        res = np.arctan2((prob * np.sin(ang)).sum(-1), (prob * np.cos(ang)).sum(-1))
    res[res < 0] += 2 * np.pi
    return res


def get_3d_bbox(scale, shift=0):
    """utils/util.py:858-886: the 8 corners [3, 8] of a box of edge lengths `scale` (scalar or 3-vector) centred at `shift`,
    in the toolkit's corner order (+y face first; x sign flips every other pair, z sign every other corner)."""
    s = np.broadcast_to(np.asarray(scale, dtype=np.float64), (3,)) / 2
    signs = np.array([[sx, sy, sz] for sy in (1, -1) for sx in (1, -1) for sz in (1, -1)], dtype=np.float64)
    return (signs * s + shift).transpose()


def transform_coordinates_3d(coordinates, RT):
    """utils/util.py:890-902: [3, N] points through a 4x4 transform (homogeneous divide)."""
    assert coordinates.shape[0] == 3
    h = RT @ np.vstack([coordinates, np.ones((1, coordinates.shape[1]), dtype=np.float32)])
    return h[:3] / h[3]


def calculate_2d_projections(coordinates_3d, intrinsics):
    """utils/util.py:905-918: pinhole projection of [3, N] camera-frame points -> int32 pixel coordinates [N, 2]."""
    p = intrinsics @ coordinates_3d
    return np.array((p[:2] / p[2]).transpose(), dtype=np.int32)


def _line(img, p, q, color, size):
    """A `size`-pixel-wide segment into an HxWx3 uint8 image (cv2.line when OpenCV is installed, a DDA with a square brush
    otherwise: the image only ever goes to a viewer)."""
    try:
        import cv2
        return cv2.line(img, tuple(int(v) for v in p), tuple(int(v) for v in q), color, size)
    except ImportError:
        pass
    p, q = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64)
    n = int(max(abs(q - p).max(), 1))
    r = max(size // 2, 0)
    for t in np.linspace(0.0, 1.0, n + 1):
        x, y = np.rint(p + t * (q - p)).astype(int)
        img[max(y - r, 0):max(y + r + 1, 0), max(x - r, 0):max(x + r + 1, 0)] = color
    return img


def draw(img, imgpts, axes, color, size=3):
    """utils/util.py:2208-2235: projected box (8 corners of get_3d_bbox) + the three axes onto `img`: bottom face darkest,
    pillars, top face in `color`, then z (blue), x (red), y (green)."""
    pts = np.int32(imgpts).reshape(-1, 2)
    ground = tuple(int(c * 0.3) for c in color)
    pillar = tuple(int(c * 0.6) for c in color)
    for i, j in zip((4, 5, 6, 7), (5, 7, 4, 6)):
        img = _line(img, pts[i], pts[j], ground, size)
    for i in range(4):
        img = _line(img, pts[i], pts[i + 4], pillar, size)
    for i, j in zip((0, 1, 2, 3), (1, 3, 0, 2)):
        img = _line(img, pts[i], pts[j], color, size)
    img = _line(img, axes[0], axes[1], (0, 0, 255), size)
    img = _line(img, axes[0], axes[3], (255, 0, 0), size)
    img = _line(img, axes[0], axes[2], (0, 255, 0), size)
    return img


def dilate_mask(mask, size=5, largest_comp=False):
    """utils/util.py:83-101: optionally keep the largest 8-connected component, fill the convex hull of the mask, dilate with a
    size x size box.  scipy.ndimage / scipy.spatial (OpenCV is not a dependency of this build); hull-edge pixels may differ from
    cv2.fillConvexPoly's rasterisation by one pixel before the dilation."""
    from scipy import ndimage
    from scipy.spatial import ConvexHull, Delaunay
    mask = mask.astype(np.uint8)
    if largest_comp:
        labels, n = ndimage.label(mask, structure=np.ones((3, 3)))
        if n:
            sizes = ndimage.sum(mask > 0, labels, index=np.arange(1, n + 1))
            mask[labels != 1 + int(np.argmax(sizes))] = 0
    ys, xs = np.where(mask)
    if len(ys) >= 3 and np.ptp(ys) > 0 and np.ptp(xs) > 0:
        pts = np.stack([xs, ys], -1).astype(np.float64)
        hull = pts[ConvexHull(pts).vertices]
        y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
        gy, gx = np.mgrid[y0:y1 + 1, x0:x1 + 1]
        inside = Delaunay(hull).find_simplex(np.stack([gx.ravel(), gy.ravel()], -1)) >= 0
        mask[y0:y1 + 1, x0:x1 + 1] |= inside.reshape(gy.shape).astype(np.uint8)
    return ndimage.binary_dilation(mask, structure=np.ones((size, size))).astype(np.uint8)


def process_data(mask_im, coord_map, inst_dict, meta_path, model_root="NOCS/obj_models"):
    """utils/util.py:2959-3067 (NOCS ground-truth parsing): instance mask image (255 = background) + NOCS coordinate map +
    {instance id: class id} + the scene's meta file -> (masks [h,w,n] uint8, coords [h,w,n,3] float32 in [0,1] with z flipped,
    class_ids [n], scales [n,3], meta words per kept instance).  Model extents are read below `model_root` (the reference
    resolves the same relative paths through hydra)."""
    cdata = np.array(mask_im, dtype=np.int32)
    instance_ids = sorted(np.unique(cdata).tolist())
    assert instance_ids[-1] == 255
    instance_ids = instance_ids[:-1]
    cdata[cdata == 255] = -1
    h, w = cdata.shape
    coord_map = np.array(coord_map, dtype=np.float32) / 255
    coord_map[:, :, 2] = 1 - coord_map[:, :, 2]
    with open(meta_path) as f:
        all_words = [line.rstrip("\n").split(" ") for line in f]
    extent = np.zeros((len(all_words), 3), dtype=np.float32)
    for i, words in enumerate(all_words):
        if len(words) == 3:                                        # real scanned objects
            if words[2].endswith("npz"):
                with np.load(os.path.join(model_root, "real_val", words[2])) as z:
                    extent[i] = z["scale"]
            else:
                extent[i] = np.loadtxt(os.path.join(model_root, "real_test", words[2] + ".txt"))
            extent[i] /= np.linalg.norm(extent[i])
        else:                                                      # CAMERA renders: bbox.txt holds two opposite corners
            path = os.path.join(model_root, "train", words[2], words[3], "bbox.txt")
            if not os.path.exists(path):
                path = os.path.join(model_root, "val", words[2], words[3], "bbox.txt")
            bbox = np.loadtxt(path)
            extent[i] = bbox[0] - bbox[1]
    for inst_id in [k for k, v in inst_dict.items() if v == 0 or k not in instance_ids]:
        del inst_dict[inst_id]
    kept = [i for i in instance_ids if i in inst_dict]
    n = len(kept)
    masks = np.zeros((h, w, n), dtype=np.uint8)
    coords = np.zeros((h, w, n, 3), dtype=np.float32)
    class_ids = np.zeros((n,), dtype=np.int_)
    scales = np.zeros((n, 3), dtype=np.float32)
    words_kept = []
    for j, inst_id in enumerate(kept):                            # instance ids are one-based
        m = cdata == inst_id
        assert m.any()
        masks[:, :, j] = m
        coords[:, :, j] = coord_map * m[..., None]
        class_ids[j] = inst_dict[inst_id]
        scales[j] = extent[inst_id - 1]
        words_kept.append(all_words[inst_id - 1])
    return masks, np.clip(coords, 0, 1), class_ids, scales, words_kept


def compute_degree_cm_mAP(final_results, synset_names, log_dir, degree_thresholds=[360], shift_thresholds=[100],
                          iou_3d_thresholds=[0.1], iou_pose_thres=0.1, use_matches_for_pose=False, num_proc=10):
    """utils/util.py:2736-2955: (iou_3d_aps [classes + 1, iou thresholds], pose_aps [classes + 1, degrees + 1, shifts + 1]) over
    the per-image result records, printed per class like the reference; the matplotlib plots it also writes into `log_dir` are
    not produced (num_proc is accepted and unused: the scorer is vectorised, not a process pool)."""
    os.makedirs(log_dir, exist_ok=True)
    iou_thr = list(iou_3d_thresholds)
    iou_aps, pose_aps = _metrics.degree_cm_mAP(final_results, synset_names, tuple(degree_thresholds), tuple(shift_thresholds),
                                               tuple(iou_thr), iou_pose_thres, use_matches_for_pose)
    degs, shifts = list(degree_thresholds) + [360], list(shift_thresholds) + [100]
    for k in (0.25, 0.5):
        if k in iou_thr:
            print("3D IoU at %d: %.1f" % (k * 100, iou_aps[-1, iou_thr.index(k)] * 100))
    for i, d in enumerate(degs):
        for j, s in enumerate(shifts):
            print("%s degree, %scm: %.1f" % (d, s, pose_aps[-1, i, j] * 100))
    return iou_aps, pose_aps
