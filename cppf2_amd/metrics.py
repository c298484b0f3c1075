"""The 5 deg / 5 cm pose criterion used to score parity: rotation/translation error between
two similarity transforms with the category symmetries of the NOCS toolkit (utils/util.py:588-663).
Host-side NumPy; the mAP machinery around it is out of scope."""
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
