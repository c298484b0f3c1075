"""NumPy restatements of the steps in front of the hot path: back-projection of a masked
depth map and the voxel down-sample.  The product path uses the HIP kernels (cppf2_amd.ops.backproject / downsample);
these stay as the checkers the tests pin to the reference, plus the pose-error helper."""
from __future__ import annotations

import numpy as np


def backproject(depth, intrinsics, instance_mask):
    """Masked depth -> points, same convention as utils/util.py:2586-2607: returns points with x and y NEGATED
    (every caller negates them back, eval.py:187-188) and the (row, col) index arrays of the used pixels."""
    depth = np.asarray(depth)
    valid = np.logical_and(instance_mask, depth > 0)
    rows, cols = np.nonzero(valid)
    z = depth[rows, cols]
    pix = np.stack([cols, rows, np.ones_like(cols)], 0).astype(np.float64)
    rays = (np.linalg.inv(intrinsics) @ pix).T
    pts = rays * z[:, None] / rays[:, -1:]
    pts[:, 0] = -pts[:, 0]
    pts[:, 1] = -pts[:, 1]
    return pts, (rows, cols)


def downsample(pc, res, rng=None):
    """One randomly chosen point per `res` voxel (utils/util.py:39-46 does this with open3d's
    voxel_down_sample_and_trace + np.random.choice); voxels are anchored at the cloud's min bound.
    Returns the indices of the kept points, ordered by voxel key."""
    pc = np.asarray(pc)
    rng = np.random if rng is None else rng
    key = np.floor((pc - pc.min(0)) / res).astype(np.int64)
    dims = key.max(0) + 1
    flat = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    order = np.argsort(flat, kind="stable")
    sf = flat[order]
    starts = np.flatnonzero(np.r_[True, sf[1:] != sf[:-1]])
    counts = np.diff(np.r_[starts, len(sf)])
    pick = starts + (rng.random_sample(len(starts)) * counts).astype(np.int64)
    return order[pick]


def rot_err_deg(R_est, R_gt, symmetric_up=False, up_axis=1):
    """Rotation error as eval.py:341-344 measures it: angle between up axes for symmetric categories, geodesic otherwise."""
    if symmetric_up:
        c = float(np.dot(R_est[:, up_axis], R_gt[:, up_axis]))
    else:
        c = (np.trace(R_est.T @ R_gt) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))
