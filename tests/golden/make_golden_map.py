"""Golden for the pose (degree / cm) mAP scorer: the reference's compute_degree_cm_mAP
(utils/util.py:2736-2955, use_matches_for_pose=False) run here on synthetic per-image result records of the on-disk
format eval.py writes (eval.py:143-147, 399).  Build container only (needs /root/reference):

    python tests/golden/make_golden_map.py
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CPPF_GOLDEN_OUT", HERE)     # tests/test_golden_regen.py regenerates into a temp dir
sys.path.insert(0, HERE)
from _ref_loader import load_reference, reference_modules  # noqa: E402


def rand_rot(rng):
    q, _ = np.linalg.qr(rng.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def small_rot(rng, deg):
    ax = rng.randn(3)
    ax /= np.linalg.norm(ax)
    a = np.radians(deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def make_results(seed=11, images=40):
    rng = np.random.RandomState(seed)
    out = []
    for im in range(images):
        n_gt = rng.randint(1, 5)
        gt_cls = rng.randint(1, 7, n_gt)
        gt_RTs, gt_scales = [], []
        for _ in range(n_gt):
            RT = np.eye(4)
            RT[:3, :3] = rand_rot(rng) * rng.uniform(0.1, 0.4)
            RT[:3, 3] = rng.randn(3) * 0.3 + np.array([0, 0, 1.0])
            gt_RTs.append(RT)
            gt_scales.append(rng.uniform(0.3, 1.0, 3))
        # predictions: every GT perturbed (some well, some badly), one dropped now and then, plus false positives
        p_cls, p_RTs, p_scales, p_scores = [], [], [], []
        for j in range(n_gt):
            if rng.rand() < 0.15:
                continue
            RT = gt_RTs[j].copy()
            deg = rng.choice([1.0, 4.0, 8.0, 14.0, 40.0])
            RT[:3, :3] = small_rot(rng, deg) @ RT[:3, :3] * rng.uniform(0.9, 1.1)
            RT[:3, 3] += rng.randn(3) * rng.choice([0.005, 0.03, 0.08])
            p_cls.append(gt_cls[j]); p_RTs.append(RT); p_scales.append(gt_scales[j] * rng.uniform(0.9, 1.1, 3))
            p_scores.append(rng.uniform(0.3, 1.0))
        for _ in range(rng.randint(0, 3)):
            RT = np.eye(4)
            RT[:3, :3] = rand_rot(rng) * 0.2
            RT[:3, 3] = rng.randn(3)
            p_cls.append(rng.randint(1, 7)); p_RTs.append(RT); p_scales.append(rng.uniform(0.3, 1.0, 3))
            p_scores.append(rng.uniform(0.0, 0.6))
        out.append(dict(gt_class_ids=gt_cls.astype(np.int32), gt_RTs=np.array(gt_RTs), gt_scales=np.array(gt_scales),
                        gt_handle_visibility=rng.randint(0, 2, n_gt).astype(np.int32),
                        pred_class_ids=np.array(p_cls, dtype=np.int32), pred_RTs=np.array(p_RTs).reshape(-1, 4, 4),
                        pred_scales=np.array(p_scales).reshape(-1, 3), pred_scores=np.array(p_scores, dtype=np.float64)))
    # (an image without ground truth makes the reference's np.stack raise, utils/util.py:2619: not generated)
    return out


def main():
    ref = load_reference()
    import copy
    results = make_results()
    names = ['BG', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']
    thr = np.linspace(0, 1, 101)
    # (the reference's worker function is pickled by name for its process pool: its modules must be importable by name there)
    with tempfile.TemporaryDirectory() as d, reference_modules():
        iou_aps, pose_aps = ref.util.compute_degree_cm_mAP(copy.deepcopy(results), names, d, degree_thresholds=[5, 10, 15],
                                                           shift_thresholds=[5, 10, 15],
                                                           iou_3d_thresholds=thr,
                                                           iou_pose_thres=0.1, use_matches_for_pose=False, num_proc=2)
    # the call eval.py:400-411 makes: pose AP over the instances matched at 3-D IoU > 0.1
    with tempfile.TemporaryDirectory() as d, reference_modules():
        iou_aps_m, pose_aps_m = ref.util.compute_degree_cm_mAP(copy.deepcopy(results), names, d, degree_thresholds=[5, 10, 15],
                                                               shift_thresholds=[5, 10, 15], iou_3d_thresholds=thr,
                                                               iou_pose_thres=0.1, use_matches_for_pose=True, num_proc=2)
    # oriented-box IoU on its own (utils/util.py:475-547 -> utils/iou.py, utils/box.py): random box pairs per class rule
    rng = np.random.RandomState(5)
    pairs = []
    for k in range(60):
        RT1, RT2 = np.eye(4), np.eye(4)
        RT1[:3, :3] = rand_rot(rng) * rng.uniform(0.5, 2.0)
        RT1[:3, 3] = rng.randn(3) * 0.1
        RT2[:3, :3] = (small_rot(rng, rng.choice([0.0, 3.0, 20.0, 70.0])) @ RT1[:3, :3]) * rng.uniform(0.8, 1.2)
        RT2[:3, 3] = RT1[:3, 3] + rng.randn(3) * rng.choice([0.0, 0.02, 0.1, 0.5])
        s1 = rng.uniform(0.2, 1.0, 3)
        s2 = s1 * rng.uniform(0.7, 1.3, 3)
        cls = names[1 + k % 6]
        hv = int(rng.randint(0, 2))
        iou = ref.util.compute_3d_iou_new(RT1.copy(), RT2.copy(), s1.copy(), s2.copy(), hv, cls, cls)
        pairs.append(dict(RT1=RT1, RT2=RT2, s1=s1, s2=s2, cls=cls, hv=hv, iou=float(iou)))
    # identical, disjoint, contained, face-touching
    I = np.eye(4)
    sh = np.eye(4); sh[:3, 3] = [2.0, 0, 0]
    half = np.eye(4); half[:3, 3] = [0.5, 0, 0]
    for RT1, RT2, s1, s2 in ((I, I, [1, 1, 1], [1, 1, 1]), (I, sh, [1, 1, 1], [1, 1, 1]), (I, I, [1, 1, 1], [.5, .4, .3]),
                             (I, half, [1, 1, 1], [1, 1, 1]), (I, half, [1.0, 2.0, 0.5], [1.0, 1.0, 1.0])):
        s1, s2 = np.array(s1, float), np.array(s2, float)
        iou = ref.util.compute_3d_iou_new(RT1.copy(), RT2.copy(), s1.copy(), s2.copy(), 1, "laptop", "laptop")
        pairs.append(dict(RT1=RT1.copy(), RT2=RT2.copy(), s1=s1, s2=s2, cls="laptop", hv=1, iou=float(iou)))
    print("pair IoUs:", [round(p_["iou"], 4) for p_ in pairs])
    with open(os.path.join(OUT, "map_results.pkl"), "wb") as f:
        pickle.dump(dict(results=results, synset_names=names, pose_aps=pose_aps, iou_aps=iou_aps, iou_thresholds=thr,
                         pose_aps_matched=pose_aps_m, iou_aps_matched=iou_aps_m, iou_pairs=pairs), f, protocol=4)
    print(pose_aps[-1])
    print(pose_aps_m[-1])
    print("IoU25 / IoU50 / IoU75 mean AP:", iou_aps[-1, 25], iou_aps[-1, 50], iou_aps[-1, 75])


if __name__ == "__main__":
    main()
