"""Pose-error helper of the evaluation report (host side).  The steps in front of the hot path (back-projection, voxel
down-sample) are HIP kernels (cppf2_amd.ops.backproject / downsample); their NumPy restatements live in
oracle/cppf_oracle.py with the rest of the test infrastructure."""
from __future__ import annotations

import numpy as np


def rot_err_deg(R_est, R_gt, symmetric_up=False, up_axis=1):
    """Rotation error as eval.py:341-344 measures it: angle between up axes for symmetric categories, geodesic otherwise."""
    if symmetric_up:
        c = float(np.dot(R_est[:, up_axis], R_gt[:, up_axis]))
    else:
        c = (np.trace(R_est.T @ R_gt) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))
