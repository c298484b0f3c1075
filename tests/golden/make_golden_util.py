"""Generates tests/golden/util_helpers.npz by RUNNING THE REAL REFERENCE helpers (through _ref_loader) that the repo-root
`utils/util.py` and `dataset.py` provide under the reference's names: real2prob / prob2real (plain + circular, torch + NumPy),
get_3d_bbox, transform_coordinates_3d, calculate_2d_projections (utils/util.py:215-272, 858-918), rotx / roty / rotz
(dataset.py:84-101).  Run in the build container only:   python tests/golden/make_golden_util.py
The output is data (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CPPF_GOLDEN_OUT", HERE)     # tests/test_golden_regen.py regenerates into a temp dir
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

ref = load_reference()
ref_dataset = ref.dataset       # the reference's module object (load_reference checks its file path)
util = ref.util


def main():
    rng = np.random.RandomState(77)
    g = {}
    vals = rng.rand(5, 7).astype(np.float32)
    g["r2p_in"] = vals
    g["r2p_np"] = util.real2prob(vals.copy(), 1.0, 32)
    g["r2p_torch"] = util.real2prob(torch.from_numpy(vals.copy()), 1.0, 32).numpy()
    ang = (rng.rand(4, 9) * 2 * np.pi).astype(np.float64)
    g["r2p_circ_in"] = ang
    g["r2p_circ_np"] = util.real2prob(ang.copy(), 2 * np.pi, 36, True)
    g["r2p_circ_torch"] = util.real2prob(torch.from_numpy(ang.copy()), 2 * np.pi, 36, True).numpy()
    prob = rng.rand(6, 32).astype(np.float32)
    prob /= prob.sum(-1, keepdims=True)
    g["p2r_in"] = prob
    g["p2r_np"] = util.prob2real(prob, 1.0, 32)
    g["p2r_torch"] = util.prob2real(torch.from_numpy(prob), 1.0, 32).numpy()
    pc = rng.rand(6, 36)
    pc /= pc.sum(-1, keepdims=True)
    g["p2r_circ_in"] = pc
    g["p2r_circ_np"] = util.prob2real(pc, 2 * np.pi, 36, True)
    g["p2r_circ_torch"] = util.prob2real(torch.from_numpy(pc), 2 * np.pi, 36, True).numpy()
    scale = np.array([0.11, 0.27, 0.09])
    g["bbox_scale"] = scale
    g["bbox_vec"] = util.get_3d_bbox(scale, 0)
    g["bbox_scalar"] = util.get_3d_bbox(0.3, 0.05)
    RT = np.eye(4)
    RT[:3, :3] = np.linalg.qr(rng.randn(3, 3))[0] * 0.7
    RT[:3, 3] = [0.05, -0.1, 0.9]
    K = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])
    g["RT"], g["K"] = RT, K
    g["bbox_cam"] = util.transform_coordinates_3d(g["bbox_vec"], RT)
    g["bbox_px"] = util.calculate_2d_projections(g["bbox_cam"], K)
    for n in ("rotx", "roty", "rotz"):
        g[n] = getattr(ref_dataset, n)(0.37)
    np.savez_compressed(os.path.join(OUT, "util_helpers.npz"), **g)
    print("wrote util_helpers.npz:", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
