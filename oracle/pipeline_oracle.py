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
                   cfg_up=(0, 1, 0), cfg_right=(1, 0, 0), cfg_front=(0, 0, 1), trig=None):
    """Sampler -> SHOT -> MLP -> decode -> votes -> pose for one scene; mirrors bench.py's GPU step."""
    n = pc.shape[0]
    idx = O.sample_tuples(seed, scene_id, num_tuples, 5, n).astype(np.int64)
    shot_feat, normal, _ = S.compute(pc, res * 10, res * 10)                   # eval.py:210
    shot_feat = np.nan_to_num(shot_feat, nan=0.0)                               # eval.py:215-216
    normal = np.nan_to_num(normal, nan=0.0)
    logits, scales = mlp_shot(weights, pc, idx, shot_feat, normal)
    if prior_fn is not None:
        logits = (logits + prior_fn(idx)).astype(np.float32)
    u = O.philox_uniform(seed, scene_id, 1, num_tuples, 6)
    out = O.run_scene(pc, idx, logits, scales, u, cfg_up, cfg_right, cfg_front, res, num_rots=num_rots, trig=trig)
    out["idx"] = idx
    return out
