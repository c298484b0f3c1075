"""BASELINE config 3 -- all six categories, DINO + SHOT ensemble voting (eval.py:84-101, 153-372) -- on the HIP path.

* full size (4096 points x 20 000 tuples x 180 rotations) on one up-symmetric category and one with the camera / mug
  axes: each model's pass (argmax, centre, back-vote mask, rotation bins, R, scale) and the ensemble selection
  (alignment losses, picked model, written RT / scale) against oracle.pipeline_oracle.run_instance_ensemble;
* all six categories through eval.main at reduced size; the checkpoint-directory layout of eval.py:91-99.
Needs an MI355X: run with `pytest -m gpu`.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cppf_oracle as O            # noqa: E402  (checker only)
from oracle import pipeline_oracle as PO       # noqa: E402  (checker only)
from cppf2_amd import synth                    # noqa: E402


@pytest.fixture()
def ev(monkeypatch):
    monkeypatch.chdir(ROOT)
    import eval as ev_
    return ev_


@pytest.mark.parametrize("category", ["bottle", "mug"])
def test_ensemble_full_size_vs_oracle(ev, category):
    N, T, R, B, seed = 4096, 20000, 180, 2, 0
    torch.manual_seed(3)
    cfg, dino_model, shot_model = ev.load_category(category)
    up_sym = category in ev.UP_SYM
    if category == "mug":
        assert cfg.right == [0, 0, 1] and cfg.front == [1, 0, 0]
    scene_ids = [10, 11]
    scenes = [synth.make_scene(seed, s, N) for s in scene_ids]
    g = torch.Generator(device="cpu").manual_seed(5)
    descs = [torch.nn.functional.normalize(torch.randn((N, 1024), generator=g), dim=-1).numpy() for _ in scenes]
    dev = torch.device("cuda")
    priors = ev._teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
    r = ev.run_ensemble(cfg, dino_model, shot_model, [s["pc"] for s in scenes], descs, seed, scene_ids, T, R,
                        opt=False, up_sym=up_sym, priors=priors, keep=True)
    pipe = r["pipe"]
    trig = (pipe.cs.cpu().numpy(), pipe.sn.cpu().numpy())
    idx_all = r["idx"].cpu().numpy()
    w_dino = {k: v.detach().cpu().numpy() for k, v in dino_model.state_dict().items()}
    w_shot = {k: v.detach().cpu().numpy() for k, v in shot_model.state_dict().items()}
    for b, scene in enumerate(scenes):
        pc = scene["pc"]
        sl = slice(b * T, (b + 1) * T)
        idx = idx_all[sl].astype(np.int64)
        assert np.array_equal(idx, O.sample_tuples(seed, scene_ids[b], T, 5, N))
        # (1) the networks: NumPy restatements of both forwards (the SHOT one on the device's descriptors) vs the device
        sub = slice(0, 2000)
        lg_d, sc_d = PO.mlp_dino(w_dino, pc, descs[b], idx[sub])
        assert np.abs(lg_d - r["kept"][0]["raw_cls"][sl][sub]).max() < 2e-4
        assert np.abs(sc_d - r["kept"][0]["pred_scales"][sl][sub]).max() < 2e-4
        ns = slice(b * N, (b + 1) * N)
        lg_s, sc_s = PO.mlp_shot(w_shot, pc, idx[sub], r["shot_feat"][ns], r["normal"][ns])
        assert np.abs(lg_s - r["kept"][1]["raw_cls"][sl][sub]).max() < 2e-4
        assert np.abs(sc_s - r["kept"][1]["pred_scales"][sl][sub]).max() < 2e-4
        # (2) both voting passes + selection: the oracle decodes the device's bins (expf ulps may flip a draw whose
        # uniform lies within 1e-5 of a CDF edge -- checked here) and gets the device's scale-head output
        per_model = []
        for m in (0, 1):
            k = r["kept"][m]
            ob = O.decode_bins(k["pred_cls"][sl], k["u"][sl], pc[idx[:, :2]], return_margin=True)
            flips = ob[0] != k["bins"][sl]
            assert np.all(ob[4][flips] < 1e-5) and flips.mean() < 1e-3
            onehot = np.full((T, 6, 32), -1e4, np.float32)
            np.put_along_axis(onehot, k["bins"][sl][:, :, None].astype(np.int64), 0.0, -1)
            per_model.append((onehot, k["pred_scales"][sl], k["u"][sl]))
        want = PO.run_instance_ensemble(pc, idx, per_model, cfg.up, cfg.right, cfg.front, cfg.res, num_rots=R,
                                        y_only=up_sym, trig=trig)
        for m in (0, 1):
            rec, o = r["records"][m][b], want["models"][m]
            assert rec["argmax"] == o["argmax"] and rec["peak"] == o["grid_obj"].max()
            assert np.array_equal(rec["t"], o["T_est"])
            assert np.array_equal(r["kept"][m]["mask"][sl], o["pairs_mask"])
            for a, name in ((0, "up"), (1, "right")):
                d = np.abs(r["kept"][m]["counts"][a, b] - o[name + "_counts"])
                # tanf ulps flip isolated cone tests: <= 4 bins off by <= 2 votes of the largest weight
                assert (d > 0).sum() <= 4 and d.max() <= 2.0 / o["imp_pair_wt"].min()
                assert int(rec[name + "_idx"]) == o[name + "_idx"]
            assert np.allclose(rec["R"], o["R_est"], atol=1e-6)
            assert np.array_equal(rec["scale"], o["pred_scale"])
            # the loss is a smooth function of R_est, which agrees to float32 rounding of the Gram-Schmidt step (1e-7)
            assert abs(r["losses"][m][b] - o["loss"]) < 1e-7
        assert int(r["pick"][b]) == want["pick"]
        assert abs(r["best"][b] - want["loss"]) < 1e-7
        assert abs(want["models"][0]["loss"] - want["models"][1]["loss"]) > 1e-5          # the selection is not a coin flip
        assert np.float32(r["scale_norm"][b]) == np.float32(want["scale_norm"])
        RT = np.eye(4)
        rec = r["records"][r["pick"][b]][b]
        RT[:3, :3], RT[:3, 3] = rec["R"] * r["scale_norm"][b], rec["t"]
        assert np.allclose(RT, want["pred_RT"], atol=1e-6)
        assert np.allclose(r["scale"][b] / r["scale_norm"][b], want["pred_scale"], atol=1e-6)
        # pose sanity vs the synthetic ground truth (teacher prior): centre within 5 mm, up axis within 5 degrees
        assert np.linalg.norm(rec["t"] - scene["t"]) < 5e-3
        assert np.degrees(np.arccos(min(1.0, float(rec["R"][:, 1] @ scene["R"][:, 1])))) < 5.0


def test_eval_main_all_six_categories(ev, tmp_path):
    rep = ev.main(num_pairs=6000, num_rots=72, num_scenes=2, num_points=1024, opt=False, out_pkl=str(tmp_path / "r.pkl"),
                  debug=True)
    assert rep["categories"] == ["bottle", "bowl", "camera", "can", "laptop", "mug"] and rep["instances"] == 12
    assert [r_["category"] for r_ in rep["results"]] == [c for c in rep["categories"] for _ in range(2)]
    assert all(r_["model"] in ("dino", "shot") and np.isfinite(r_["loss"]) for r_ in rep["results"])
    assert set(rep["acc_5deg_5cm_per_category"]) == set(rep["categories"])
    # up-symmetric categories are scored on the up axis only; all of them recover the teacher's pose
    for c in ("bottle", "bowl", "can"):
        assert rep["acc_5deg_5cm_per_category"][c] >= 0.5
    import pickle
    rec = pickle.load(open(tmp_path / "r.pkl", "rb"))
    assert rec["pred_RTs"].shape == (12, 4, 4) and rec["pred_class_ids"].tolist() == [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6]
    from cppf2_amd import metrics
    _, aps = metrics.degree_cm_mAP([rec], use_matches_for_pose=True)
    for c in rep["categories"]:
        assert abs(aps[ev.category2id[c], 2, 2] - rep["pose_AP_per_category"][c]["15deg_15cm"]) < 1e-12
    rep2 = ev.main(num_pairs=3000, num_rots=36, num_scenes=1, num_points=512, opt=False, categories="mug,can",
                   visual_branch=False, debug=True)
    assert rep2["categories"] == ["mug", "can"] and [r_["model"] for r_ in rep2["results"]] == ["dino", "dino"]
    # with the online refinement (eval.py:319-355, `opt`; one kernel per batch): poses stay finite and do not get worse
    rep3 = ev.main(num_pairs=6000, num_rots=72, num_scenes=2, num_points=1024, opt=True, debug=True)
    assert rep3["instances"] == 12 and "Adam" in rep3["opt_refinement"]
    for a, b in zip(rep["results"], rep3["results"]):
        assert np.all(np.isfinite(np.array(b["pred_RT"]))) and b["category"] == a["category"]
        # (the y-only objective of an up-symmetric category leaves the translation along the axis free: within 2 cm)
        assert b["tr_err_cm"] < 2.0 and b["rot_err_deg"] < a["rot_err_deg"] + 2.0
    assert rep3["acc_5deg_5cm"] >= rep["acc_5deg_5cm"]
    # teacher box extents stand in for the scale head -> the 3-D IoU matching of the scorer sees real boxes
    assert rep["iou_AP"]["IoU50"] >= 0.9 and rep3["iou_AP"]["IoU50"] >= 0.9
    assert rep["pose_AP"]["10deg_5cm"] >= 0.9 and rep3["pose_AP"]["5deg_5cm"] >= rep["pose_AP"]["5deg_5cm"]


def test_checkpoint_directory_layout(ev, tmp_path):
    """eval.py:91-99: <ckpt_dir>/{dino,shot}/<cat>-num_more-3/{.hydra/config.yaml, lightning_logs/version_0/checkpoints/
    last.ckpt}; each model is built from its own saved cfg and the loop keeps the SHOT run's cfg (eval.py:101)."""
    import yaml
    from cppf2_amd.models import BeyondCPPFDino, BeyondCPPFShot

    class C3:
        num_more = 3
    saved = {}
    for name, cls, res in (("dino", BeyondCPPFDino, 2e-3), ("shot", BeyondCPPFShot, 3e-3)):
        root = tmp_path / name / "mug-num_more-3"
        (root / ".hydra").mkdir(parents=True)
        (root / "lightning_logs" / "version_0" / "checkpoints").mkdir(parents=True)
        with open(root / ".hydra" / "config.yaml", "w") as f:      # a resolved hydra config, as the trainer leaves it
            yaml.safe_dump(dict(res=res, max_epoch=200, opt=dict(weight_decay=0, lr=1e-3), up=[0, 1, 0], right=[0, 0, 1],
                                front=[1, 0, 0], num_more=3, category=6, cat_name="mug"), f)
        torch.manual_seed(40 + len(saved))
        m = cls(C3())
        torch.save({"state_dict": m.state_dict()}, root / "lightning_logs" / "version_0" / "checkpoints" / "last.ckpt")
        saved[name] = {k: v.clone() for k, v in m.state_dict().items()}
    cfg, dino_model, shot_model = ev.load_category("mug", ckpt_dir=str(tmp_path))
    assert cfg.res == 3e-3 and cfg.right == [0, 0, 1] and cfg.front == [1, 0, 0] and cfg.num_more == 3
    for model, name in ((dino_model, "dino"), (shot_model, "shot")):
        sd = model.state_dict()
        assert all(torch.equal(sd[k].cpu(), v) for k, v in saved[name].items())
    # a category without a checkpoint directory falls back to config/ (random-init weights)
    cfg2, _, _ = ev.load_category("bottle", ckpt_dir=str(tmp_path))
    assert cfg2.res == 2e-3 and cfg2.up_sym is True
    rep = ev.main(num_pairs=3000, num_rots=36, num_scenes=1, num_points=512, opt=False, category="mug",
                  ckpt_dir=str(tmp_path), debug=True)
    assert rep["instances"] == 1 and np.all(np.isfinite(np.array(rep["results"][0]["pred_RT"])))


def test_run_ensemble_is_the_same_with_the_bin_draw_fused_into_the_mlp(ev):
    """eval.run_ensemble with keep=False draws the bins inside the logit heads' output layers (and gathers the SHOT model's
    tuple rows inside its first layer); keep=True goes through the materialised logits: identical records and selection."""
    from cppf2_amd import synth
    dev = torch.device("cuda:0")
    cfg, dino, shot_m = ev.load_category("mug", device=dev)
    B = 3
    scenes = [synth.make_scene(3, 40 + s, 1024) for s in range(B)]
    g = torch.Generator().manual_seed(2)
    descs = [torch.nn.functional.normalize(torch.randn((1024, 1024), generator=g), dim=-1).numpy() for _ in scenes]
    prior = ev._teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
    kw = dict(priors=prior, scale_priors=np.stack([s["extent"] for s in scenes]))
    a = ev.run_ensemble(cfg, dino, shot_m, [s["pc"] for s in scenes], descs, 5, [40, 41, 42], 6000, 72, keep=True, **kw)
    b = ev.run_ensemble(cfg, dino, shot_m, [s["pc"] for s in scenes], descs, 5, [40, 41, 42], 6000, 72, keep=False, **kw)
    for m in (0, 1):
        assert a["records"][m].tobytes() == b["records"][m].tobytes()
    assert np.array_equal(a["losses"], b["losses"]) and np.array_equal(a["pick"], b["pick"])


@pytest.mark.parametrize("cfg_seed", [0, 1, 2])
def test_run_ensemble_ragged_random_sizes_fused_equals_materialised(ev, cfg_seed):
    """The product path (rows gathered / tables summed inside the first layers, bins drawn in the output layers' epilogues, scale
    head on the kept pairs, two streams) against the materialised one-stream form (keep=True) on batches drawn from a seed: 1-5
    instances of 150-5000 points each (ragged), 700-9000 pairs (not a multiple of any tile), 24-180 rotations, a random category,
    refinement on or off: records, losses and picks byte for byte."""
    rng = np.random.RandomState(4000 + cfg_seed)
    dev = torch.device("cuda:0")
    cat = ["bottle", "bowl", "camera", "can", "laptop", "mug"][int(rng.randint(6))]
    cfg, dino, shot_m = ev.load_category(cat, device=dev)
    B = int(rng.randint(1, 6))
    Ns = [int(rng.randint(150, 5000)) for _ in range(B)]
    T = int(rng.randint(700, 9000))
    R = int(rng.choice([24, 45, 72, 97, 180]))
    ids = [int(x) for x in rng.randint(0, 1000, size=B)]
    scenes = [synth.make_scene(7 + cfg_seed, i, n) for i, n in zip(ids, Ns)]
    g = torch.Generator().manual_seed(cfg_seed)
    descs = [torch.nn.functional.normalize(torch.randn((n, 1024), generator=g), dim=-1).numpy() for n in Ns]
    prior = ev._teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
    kw = dict(priors=prior, scale_priors=np.stack([s["extent"] for s in scenes]), opt=bool(rng.randint(2)), up_sym=cat in ev.UP_SYM)
    a = ev.run_ensemble(cfg, dino, shot_m, [s["pc"] for s in scenes], descs, 11, ids, T, R, keep=True, **kw)
    b = ev.run_ensemble(cfg, dino, shot_m, [s["pc"] for s in scenes], descs, 11, ids, T, R, keep=False, **kw)
    for m in (0, 1):
        assert a["records"][m].tobytes() == b["records"][m].tobytes(), (cat, B, Ns, T, R, m)
    assert np.array_equal(a["losses"], b["losses"], equal_nan=True) and np.array_equal(a["pick"], b["pick"])


def test_run_ensemble_on_two_streams_equals_the_one_stream_order(ev):
    """The product's batch mode (DINO pass and SHOT pass on two HIP streams, twin pipelines) writes the same records, losses and
    picks as the one-stream order -- three times in a row (the streams and their scratch buffers are reused)."""
    N, T, R, seed = 2048, 12000, 90, 1
    torch.manual_seed(5)
    cfg, dino_model, shot_model = ev.load_category("mug")
    scene_ids = [3, 4, 5]
    scenes = [synth.make_scene(seed, s_, N) for s_ in scene_ids]
    g = torch.Generator(device="cpu").manual_seed(9)
    descs = [torch.nn.functional.normalize(torch.randn((N, 1024), generator=g), dim=-1).numpy() for _ in scenes]
    priors = ev._teacher_prior(np.concatenate([s_["pc_canon"] for s_ in scenes]), torch.device("cuda"))
    kw = dict(opt=True, up_sym=False, priors=priors, scale_priors=np.stack([s_["extent"] for s_ in scenes]))
    a = ev.run_ensemble(cfg, dino_model, shot_model, [s_["pc"] for s_ in scenes], descs, seed, scene_ids, T, R, two_streams=False, **kw)
    for _ in range(3):
        b = ev.run_ensemble(cfg, dino_model, shot_model, [s_["pc"] for s_ in scenes], descs, seed, scene_ids, T, R, two_streams=True, **kw)
        for m in (0, 1):
            assert a["records"][m].tobytes() == b["records"][m].tobytes()
        assert a["selected"].tobytes() == b["selected"].tobytes()
        assert np.array_equal(a["losses"], b["losses"]) and np.array_equal(a["pick"], b["pick"])
