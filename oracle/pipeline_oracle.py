"""Whole-scene CPU oracle: eval.py:207-313 for one SHOT-model scene, MLP included (NumPy matmuls).
TEST INFRASTRUCTURE, NOT PRODUCT: used by tests and by bench.py's cpu_baseline leg only."""
import numpy as np

from . import cppf_oracle as O
from . import shot_oracle as S


def _res_layer(x, w, prefix):
    """train_shot.py:40-44: fc2(relu(fc1(x))) + (fc0(x) | x)."""
    h = np.maximum(x @ w[prefix + "fc1.weight"].T + w[prefix + "fc1.bias"], 0)
    h = h @ w[prefix + "fc2.weight"].T + w[prefix + "fc2.bias"]
    if prefix + "fc0.weight" in w:
        x = x @ w[prefix + "fc0.weight"].T + w[prefix + "fc0.bias"]
    return (h + x).astype(np.float32)


def _stack(x, w, name):
    i = 0
    while "%s.%d.fc1.weight" % (name, i) in w:
        x = _res_layer(x, w, "%s.%d." % (name, i))
        i += 1
    return x


def mlp_shot(weights, pc, idx, shot_feat, normal):
    """BeyondCPPF(SHOT).forward (train_shot.py:117-122) in NumPy float32."""
    feat = _stack(shot_feat.astype(np.float32), weights, "shot_encoder")
    x = O.prepare_tuple_inputs_shot(pc, idx, feat, normal)
    f = _stack(x, weights, "tuple_encoder")
    scales = _stack(f, weights, "scale_encoder")
    logits = _stack(f, weights, "logit_encoder").reshape(f.shape[0], 6, -1)
    return logits, scales


def run_scene_full(weights, pc, seed, scene_id, num_tuples, res=2e-3, num_rots=180, prior_fn=None,
                   cfg_up=(0, 1, 0), cfg_right=(1, 0, 0), cfg_front=(0, 0, 1), trig=None, shot_threads=1, topk_impl="numpy",
                   topk_threads=0, timings=None):
    """Sampler -> SHOT -> MLP -> decode -> votes -> pose for one scene; mirrors bench.py's GPU step.
    shot_threads / topk_impl / topk_threads: how the descriptor and the get_topk_dir stages run (results identical for every
    setting); timings: a dict that receives seconds per stage."""
    import time
    t_ = [time.perf_counter()]

    def lap(name):
        t_.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + t_[-1] - t_[-2]
    n = pc.shape[0]
    idx = O.sample_tuples(seed, scene_id, num_tuples, 5, n).astype(np.int64)
    # eval.py:210; the normals in pcl::NormalEstimation's arithmetic, like the product's default (cppf2_amd.shot.ARITHMETIC)
    lap("sample_tuples")
    shot_feat, normal, _, _ = S.compute_ex(pc, res * 10, res * 10, pcl_arithmetic=True, threads=shot_threads)
    shot_feat = np.nan_to_num(shot_feat, nan=0.0)                               # eval.py:215-216
    normal = np.nan_to_num(normal, nan=0.0)
    lap("shot_descriptor")
    logits, scales = mlp_shot(weights, pc, idx, shot_feat, normal)
    lap("mlp")
    if prior_fn is not None:
        logits = (logits + prior_fn(idx)).astype(np.float32)
    u = O.philox_uniform(seed, scene_id, 1, num_tuples, 6)
    lap("prior_and_uniforms (bench setup, not the path)")
    out = O.run_scene(pc, idx, logits, scales, u, cfg_up, cfg_right, cfg_front, res, num_rots=num_rots, trig=trig,
                      topk_impl=topk_impl, topk_threads=topk_threads)
    lap("decode_votes_backvote_rotation_bins_pose")
    out["idx"] = idx
    return out


def mlp_dino(weights, pc, desc, idx):
    """BeyondCPPF(DINO).forward (train_dino.py:91-97, 128-133) in NumPy float32, in the reference's order: gather the
    1024-d descriptors of the k tuple points, desc_transform each, concatenate, desc_pair_transform."""
    idx = np.asarray(idx).astype(np.int64)
    k = idx.shape[1]
    desc = np.asarray(desc, dtype=np.float32)
    wt, bt = weights["desc_transform.weight"], weights["desc_transform.bias"]
    parts = [(desc[idx[:, i]] @ wt.T + bt).astype(np.float32) for i in range(k)]                 # train_dino.py:95
    dp = (np.concatenate(parts, -1) @ weights["desc_pair_transform.weight"].T
          + weights["desc_pair_transform.bias"]).astype(np.float32)                              # train_dino.py:96
    x = np.concatenate([O.tuple_coord_inputs(pc, idx), dp], -1).astype(np.float32)
    f = _stack(x, weights, "tuple_encoder")
    scales = _stack(f, weights, "scale_encoder")
    logits = _stack(f, weights, "logit_encoder").reshape(f.shape[0], 6, -1)
    return logits, scales


def alignment_loss(pc, T_est, R_est, pred_scale_norm, idx_filtered, pred_pairs_filtered, y_only):
    """eval.py:358-363: clipped L1 between the canonicalised points of the kept pairs and their decoded (un-scaled)
    coordinates; float32 points minus a float64 centre -> float64 throughout."""
    pc_canon = (np.asarray(pc, dtype=np.float32) - T_est) @ R_est / pred_scale_norm             # eval.py:358
    loss = np.abs(pc_canon[idx_filtered[:, :2]] - pred_pairs_filtered)                           # eval.py:359
    if y_only:
        loss = loss[..., 1]                                                                      # eval.py:360-361
    loss = np.clip(loss, 0, 0.1)                                                                 # eval.py:362
    return loss.mean()                                                                           # eval.py:363


def run_instance_ensemble(pc, idx, per_model, cfg_up, cfg_right, cfg_front, res, num_rots=180, y_only=False,
                          geo_branch=True, visual_branch=True, trig=None, angle_tol=1.0, backproj_ratio=0.1,
                          imp_wt_margin=0.01, topk_impl="numpy", topk_threads=0):
    """eval.py:217-372 for one instance (opt off): both models vote, each pose is scored by the alignment loss, the
    smaller one wins.  per_model: [(pred_cls, pred_scales, uniforms) for the DINO model, ... for the SHOT model] -- the
    networks' outputs are inputs here (mlp_dino / mlp_shot produce them).  Reproduced quirks: the scale (median of the
    kept pairs' scale head) is taken from model 0 only and reused for model 1 (eval.py:308-310); geo_branch gates
    model 0 and visual_branch model 1 (eval.py:367); strict '<' against best_loss = inf."""
    best_loss, best_idx = np.inf, -1
    pred_scale = pred_scale_norm = None
    outs = []
    for model_idx, (pred_cls, pred_scales, uniforms) in enumerate(per_model):
        o = O.run_scene(pc, idx, pred_cls, pred_scales, uniforms, cfg_up, cfg_right, cfg_front, res, num_rots=num_rots,
                        angle_tol=angle_tol, backproj_ratio=backproj_ratio, imp_wt_margin=imp_wt_margin, trig=trig,
                        topk_impl=topk_impl, topk_threads=topk_threads)
        if model_idx == 0:
            pred_scale = o["pred_scale"]                                                         # eval.py:309
            pred_scale_norm = np.linalg.norm(pred_scale)                                         # eval.py:310 (float32)
        mask = o["pairs_mask"]
        o["loss"] = alignment_loss(pc, o["T_est"], o["R_est"], pred_scale_norm, np.asarray(idx)[mask],
                                   o["pred_pairs"][mask], y_only)
        if o["loss"] < best_loss and ((geo_branch and model_idx == 0) or (visual_branch and model_idx == 1)):
            best_loss, best_idx = o["loss"], model_idx                                           # eval.py:367-369
        outs.append(o)
    RT = np.eye(4)
    scale = np.ones(3)
    if best_idx >= 0:
        RT[:3, :3] = outs[best_idx]["R_est"] * pred_scale_norm                                   # eval.py:370
        RT[:3, 3] = outs[best_idx]["T_est"]                                                      # eval.py:371
        scale = pred_scale / pred_scale_norm                                                     # eval.py:372
    return dict(models=outs, pick=best_idx, loss=best_loss, pred_RT=RT, pred_scale=scale, scale_norm=pred_scale_norm)
