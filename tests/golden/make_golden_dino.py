"""Golden vectors for the DINO-branch feature plumbing: the reference's own interpolate_features
(dataset.py:40-59) run here on CPU torch.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_dino.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CPPF_GOLDEN_OUT", HERE)     # tests/test_golden_regen.py regenerates into a temp dir
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402


def main():
    ref_dataset = load_reference().dataset
    rng = np.random.RandomState(7)
    C, h, w, stride = 48, 9, 13, 4
    desc = rng.randn(1, C, h, w).astype(np.float32)
    n = 300
    pts = np.stack([rng.uniform(-3, w * stride + 3, n), rng.uniform(-3, h * stride + 3, n)], -1).astype(np.float32)
    # exact pixel centres, image corners and far-outside points
    pts[:8] = np.array([[0, 0], [w * stride - 1, h * stride - 1], [1.5, 1.5], [5.5, 9.5], [-40, 3], [3, 400],
                        [w * stride - 0.5, 0.0], [2.0, h * stride - 0.5]], np.float32)
    out = {}
    for name, norm in (("normalized", True), ("raw", False)):
        r = ref_dataset.interpolate_features(torch.from_numpy(desc), torch.from_numpy(pts)[None], strides=stride,
                                             normalize=norm)
        out[name] = r[0].T.contiguous().numpy()          # [n, C], as DINOV2.forward returns it (dataset.py:79-80)
    np.savez_compressed(os.path.join(OUT, "dino_interp.npz"), desc=desc, pts=pts, stride=np.int32(stride), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
