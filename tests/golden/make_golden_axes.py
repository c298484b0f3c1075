"""Generates tests/golden/axes.npz by RUNNING THE REAL REFERENCE functions (through _ref_loader) with the axis
convention of the `camera` / `mug` categories (config/category/camera.yaml:5-6: front = [1,0,0], right = [0,0,1];
up stays [0,1,0]) -- the only non-default axes the reference ships -- next to the default ones, plus

  * a11 pose assembly: eval.py:295-313 is inline in eval.main, so the same NumPy statements are executed here on the
    reference's own top-1 directions (line cited next to each statement) for both axis conventions;
  * a4 decode: torch.softmax of seeded [T,6,32] logits (eval.py:228), the probabilities torch.multinomial
    (eval.py:230, unseeded) draws from -- pins the oracle's normalised CDF.

Run in the build container only:   python tests/golden/make_golden_axes.py
The output is data (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CPPF_GOLDEN_OUT", HERE)     # tests/test_golden_regen.py regenerates into a temp dir
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (imports the reference through _ref_loader; main() is not run)
from cppf2_amd import synth  # noqa: E402

ref = mg.ref
t = mg.t

AXES = {"default": ([0, 1, 0], [1, 0, 0], [0, 0, 1]),            # config/config.yaml:12-14  (up, right, front)
        "camera": ([0, 1, 0], [0, 0, 1], [1, 0, 0])}             # config/category/camera.yaml:5-6, mug.yaml:5-6


def assemble_reference(preds_up, preds_right, cfg_up, cfg_right):
    """eval.py:295-313 (model_idx-independent part), statement for statement."""
    preds_up = np.array(preds_up)
    preds_right = np.array(preds_right)
    preds_right -= np.dot(preds_up, preds_right) * preds_up                          # eval.py:295
    preds_right /= (np.linalg.norm(preds_right) + 1e-9)                              # eval.py:296
    up_loc = np.where(cfg_up)[0][0]                                                  # eval.py:298
    right_loc = np.where(cfg_right)[0][0]                                            # eval.py:299
    R_est = np.eye(3)                                                                # eval.py:300
    R_est[:3, up_loc] = preds_up                                                     # eval.py:301
    R_est[:3, right_loc] = preds_right                                               # eval.py:302
    other_loc = list(set([0, 1, 2]) - set([up_loc, right_loc]))[0]                   # eval.py:312
    R_est[:3, other_loc] = np.cross(R_est[:3, (other_loc + 1) % 3], R_est[:3, (other_loc + 2) % 3])   # eval.py:313
    return R_est, preds_right


def main():
    sphere_pts = np.array(ref.fibonacci_sphere(720), dtype=np.float32)               # eval.py:79-80
    g = dict(sphere_pts=sphere_pts)
    rng = np.random.RandomState(4321)
    N, T, R = 256, 512, 36
    scene = synth.make_scene(11, 2, n_points=N)
    pc = scene["pc"]
    idx = rng.randint(0, N, (T, 5)).astype(np.int64)
    scaled = mg.synth_votes(scene, idx, rng)
    cs, sn = mg.torch_trig(R)
    g.update(pc=pc, idx=idx, scaled=scaled, cos=cs, sin=sn, R_gt=scene["R"], t_gt=scene["t"])
    for name, (up, right, front) in AXES.items():
        mg.UP, mg.RIGHT, mg.FRONT = up, right, front
        out = mg.run_reference_scene(pc, idx, scaled, R, sphere_pts)
        for k, v in out.items():
            g["%s_%s" % (name, k)] = v
        R_est, right_orth = assemble_reference(out["up_top5_dirs"][0], out["right_top5_dirs"][0],
                                               np.array(up), np.array(right))
        g["%s_axes" % name] = np.array([up, right, front], dtype=np.int64)
        g["%s_R_est" % name] = R_est
        g["%s_right_orth" % name] = right_orth

    # ensemble score, eval.py:358-363 (inline in eval.main): the same statements on the reference's own T_est / R_est /
    # pairs_mask of the two axis conventions, decoded coordinates drawn at random (their values do not matter here)
    pp = (rng.randint(0, 32, (T, 2, 3)).astype(np.float32) / np.float32(31) - np.float32(0.5))
    pred_scale = np.array([0.31, 0.52, 0.29], dtype=np.float32)
    pred_scale_norm = np.linalg.norm(pred_scale)                                     # eval.py:310
    g.update(loss_pred_pairs=pp, loss_pred_scale=pred_scale)
    for name in AXES:
        T_est, R_est, pairs_mask = g[name + "_T_est"], g[name + "_R_est"], g[name + "_pairs_mask"]
        point_idxs_all_filtered = idx[pairs_mask]                                    # eval.py:267
        for y_only in (False, True):
            pc_canon = (pc - T_est) @ R_est / pred_scale_norm                        # eval.py:358
            loss = np.abs(pc_canon[point_idxs_all_filtered[:, :2]] - pp[pairs_mask])  # eval.py:359
            if y_only:
                loss = loss[..., 1]                                                  # eval.py:360-361
            loss = np.clip(loss, 0, 0.1)                                             # eval.py:362
            g["%s_loss_%s" % (name, "y" if y_only else "xyz")] = np.float64(loss.mean())   # eval.py:363

    # a4: softmax of seeded logits (eval.py:226-228); float32 like the model output
    tg = torch.Generator().manual_seed(77)
    logits = (torch.randn((64, 6, 32), generator=tg) * 3.0)
    logits[0, 0] = 0.0                      # uniform row
    logits[1, 0, 5] = 60.0                  # one-hot after exp underflow of the rest
    logits[2, 0] = -80.0                    # all equal, large negative
    prob = torch.softmax(logits, -1)
    g.update(softmax_logits=logits.numpy(), softmax_prob=prob.numpy())
    np.savez_compressed(os.path.join(OUT, "axes.npz"), **g)
    print({k: getattr(v, "shape", None) for k, v in g.items()})
    for name in AXES:
        print(name, "T_est", g[name + "_T_est"], "up", g[name + "_up_top1"], "right", g[name + "_right_top1"])
        print(g[name + "_R_est"])
    print("R_gt", scene["R"])


if __name__ == "__main__":
    main()
