"""CPU tests of host-side logic: config surface, soft-bin targets vs the reference golden table, geometry helpers,
model state-dict layout vs the reference checkpoint layout, synthetic scene invariants."""
import os
import sys

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_surface():
    from cppf2_amd.config import load_config
    c = load_config(os.path.join(ROOT, "config"), "config", ["category=bottle", "opt.lr=5e-4"])
    assert c.res == 2e-3 and isinstance(c.res, float)
    assert c.cat_name == "bottle" and c.category == 1 and c.up_sym is True
    assert c.up == [0, 1, 0] and c.right == [1, 0, 0] and c.front == [0, 0, 1] and c.num_more == 3
    assert c.opt.lr == 5e-4 and c.opt.weight_decay == 0
    c = load_config(os.path.join(ROOT, "config"), "config")
    assert c.cat_name == "bowl"                       # defaults: - category: bowl
    for name in ("bottle", "bowl", "camera", "can", "laptop", "mug"):
        assert load_config(os.path.join(ROOT, "config"), "config", ["category=" + name]).cat_name == name
    c = load_config(os.path.join(ROOT, "config"), "custom")
    assert c.res == 2e-3 and "cat_name" not in c


def test_real2prob_golden(small):
    from cppf2_amd.training import real2prob
    got = real2prob(torch.from_numpy(small["r2p_vals"]), 1.0, 32).numpy()
    assert np.array_equal(got, small["r2p_table"])


def test_model_state_dict_layout_and_forward_shapes():
    # pure-torch parts only (the HIP encode needs a GPU): key names/shapes = the reference's checkpoint layout
    from cppf2_amd.models import BeyondCPPFDino, BeyondCPPFShot

    class Cfg:
        num_more = 3
    g = np.load(os.path.join(GOLDEN, "model_shot.npz"))
    m = BeyondCPPFShot(Cfg())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    # shot_encoder + heads against the reference forward, feeding the reference's own tuple inputs layout
    from oracle import cppf_oracle as O
    with torch.no_grad():
        feat = m.shot_encoder(torch.from_numpy(g["shot_raw"]))
        x = O.prepare_tuple_inputs_shot(g["pc"], g["idx"], feat.numpy(), g["normal"])
        cls, sc = m.heads(torch.from_numpy(x))
    assert np.allclose(cls.numpy(), g["pred_cls"], atol=2e-5) and np.allclose(sc.numpy(), g["pred_scales"], atol=2e-5)
    gd = np.load(os.path.join(GOLDEN, "model_dino.npz"))
    torch.manual_seed(int(gd["seed"]))
    md = BeyondCPPFDino(Cfg())
    assert list(md.state_dict().keys()) == list(gd["keys"])
    assert [str(tuple(v.shape)) for v in md.state_dict().values()] == list(gd["shapes"])


def test_backproject_and_downsample():
    from oracle import cppf_oracle as O
    K = np.array([[500.0, 0, 32], [0, 500.0, 24], [0, 0, 1]])
    depth = np.zeros((48, 64))
    depth[10:30, 20:50] = 0.8
    mask = np.zeros_like(depth, bool)
    mask[5:25, 25:60] = True
    pts, (rows, cols) = O.backproject(depth, K, mask)
    assert pts.shape == (15 * 25, 3) and rows.min() == 10 and cols.max() == 49
    assert np.allclose(pts[:, 2], 0.8)
    # x,y are negated (utils/util.py:2604-2605): pixel right of the principal point -> negative x
    j = np.argmax(cols)
    assert pts[j, 0] < 0 and np.isclose(-pts[j, 0], (cols[j] - 32) / 500.0 * 0.8)
    keep = O.downsample(pts, 0.01, np.random.RandomState(0))
    vox = np.floor((pts[keep] - pts.min(0)) / 0.01).astype(int)
    assert len(np.unique(vox, axis=0)) == len(keep)                       # one point per voxel
    assert len(keep) == len(np.unique(np.floor((pts - pts.min(0)) / 0.01).astype(int), axis=0))


def test_synthetic_scene_invariants():
    from cppf2_amd import synth
    a, b = synth.make_scene(3, 5, 512), synth.make_scene(3, 5, 512)
    assert np.array_equal(a["pc"], b["pc"])                               # seeded, machine independent
    assert np.abs(a["pc_canon"]).max() <= 0.5 + 1e-6
    back = a["pc_canon"].astype(np.float64) * a["diag"] @ a["R"].T + a["t"]
    assert np.abs(back - a["pc"]).max() < 1e-6
    assert np.allclose(a["R"] @ a["R"].T, np.eye(3), atol=1e-12)
    idx = synth.host_sample_tuples(3, 5, 100, 5, 512)
    from oracle import cppf_oracle as O
    assert np.array_equal(idx, O.sample_tuples(3, 5, 100, 5, 512))
    assert np.array_equal(synth.uniforms(3, 5, 1, 50, 6).astype(np.float32), O.philox_uniform(3, 5, 1, 50, 6))


def test_5deg5cm_criterion_golden():
    """cppf2_amd.metrics.rt_degree_cm against utils/util.py:588-663 run on seeded poses (tests/golden/metric_5deg5cm.npz)."""
    from cppf2_amd.metrics import SYNSET_NAMES, rt_degree_cm
    g = np.load(os.path.join(GOLDEN, "metric_5deg5cm.npz"))
    for i, cid, hv, theta, shift in g["table"]:
        got = rt_degree_cm(g["A"][int(i)], g["B"][int(i)], SYNSET_NAMES[int(cid)], int(hv))
        assert np.isclose(got[0], theta, atol=1e-9) and np.isclose(got[1], shift, atol=1e-12)


def test_bin_lut_is_conservative():
    """Every (direction, bin) pair inside the cone must be listed in the direction's cell -- for the fibonacci bins
    and for an arbitrary bin set -- and a wide cone makes the builder decline (exhaustive kernel is used then)."""
    from cppf2_amd import ops
    rng = np.random.RandomState(0)
    for sph, thr in ((ops.sphere_bins(1.0), ops.cone_threshold(1.0)),
                     (rng.randn(300, 3).astype(np.float32), ops.cone_threshold(1.0))):
        sph = sph / np.linalg.norm(sph, axis=1, keepdims=True)
        lut = ops.build_bin_lut(sph, thr)
        assert lut is not None and lut.shape == (ops.LUT_ROWS, ops.LUT_COLS, ops.LUT_K)
        v = rng.randn(200000, 3)
        v[:2000] = sph[rng.randint(0, len(sph), 2000)] + rng.randn(2000, 3) * 0.01     # many directions near bins
        v[2000:2100] = [0, 1, 0] + rng.randn(100, 3) * 0.02                              # near the poles
        v[2100:2200] = [0, -1, 0] + rng.randn(100, 3) * 0.02
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        hit_v, hit_s = np.nonzero(v @ sph.T.astype(np.float32) > np.float32(thr))
        phi = np.arctan2(v[:, 2], v[:, 0]).astype(np.float32)
        phi = np.where(phi < 0, phi + np.float32(2 * np.pi), phi)
        ci = np.clip(((1 - v[:, 1]) * np.float32(ops.LUT_ROWS / 2)).astype(int), 0, ops.LUT_ROWS - 1)
        cj = np.clip((phi * np.float32(ops.LUT_COLS / (2 * np.pi))).astype(int), 0, ops.LUT_COLS - 1)
        listed = (lut[ci[hit_v], cj[hit_v]] == hit_s[:, None]).any(1)
        assert len(hit_v) > 1000 and listed.all()
    assert ops.build_bin_lut(ops.sphere_bins(1.0), np.cos(np.radians(30))) is None


def test_example_data_backproject_golden(full_summary):
    """BASELINE config 1 plumbing: back-projection of the reference's example depth+mask equals the reference's
    utils/util.py:2586 output bit for bit; voxel down-sample gives the 4 251 voxels SURVEY.md reports."""
    import hashlib
    from PIL import Image
    from oracle import cppf_oracle as O
    e = full_summary["example_backproject"]
    ex = os.path.join(GOLDEN, "example_data")
    depth = np.array(Image.open(os.path.join(ex, "depth.png"))).astype(np.float64) / e["depth_scale"]
    mask = np.array(Image.open(os.path.join(ex, "mask.png")))
    mask = (mask[..., 0] if mask.ndim == 3 else mask) > 0
    pts, (rows, cols) = O.backproject(depth, np.array(e["K"]), mask)
    assert pts.shape[0] == e["n"]
    assert hashlib.sha256(np.ascontiguousarray(pts).tobytes()).hexdigest() == e["sha"]
    assert hashlib.sha256(np.ascontiguousarray(np.stack([rows, cols], -1)).tobytes()).hexdigest() == e["rows_sha"]
    pc = pts.copy()
    pc[:, :2] = -pc[:, :2]
    keep = O.downsample(pc.astype(np.float32), 2e-3, np.random.RandomState(0))
    assert len(keep) == 4251


def test_pose_map_scorer_matches_the_reference_toolkit():
    """degree/cm pose AP == the reference's compute_degree_cm_mAP (utils/util.py:2736-2955) on 40
    synthetic images in the on-disk record format (golden: tests/golden/make_golden_map.py)."""
    import pickle
    from cppf2_amd import metrics
    with open(os.path.join(GOLDEN, "map_results.pkl"), "rb") as f:
        g = pickle.load(f)
    aps = metrics.pose_mAP(g["results"], g["synset_names"], (5, 10, 15), (5, 10, 15))
    assert aps.shape == g["pose_aps"].shape == (8, 4, 4)
    assert np.allclose(aps, g["pose_aps"], rtol=0, atol=1e-12, equal_nan=True)
    assert 0.05 < aps[-1, 0, 0] < aps[-1, -1, -1] <= 1.0
    # the record builder writes what the scorer (and the reference's pickles) expect
    r0 = g["results"][0]
    rec = metrics.make_result_record(r0["pred_class_ids"], r0["pred_RTs"], r0["pred_scales"], r0["pred_scores"],
                                     r0["gt_class_ids"], r0["gt_RTs"], r0["gt_scales"], r0["gt_handle_visibility"])
    assert set(metrics.RESULT_KEYS) <= set(rec) and rec["pred_RTs"].dtype == np.float64
    one = metrics.pose_mAP([rec], g["synset_names"])
    assert np.allclose(one, metrics.pose_mAP([r0], g["synset_names"]), equal_nan=True)
    # perfect predictions score 1 for every class that occurs
    perfect = [metrics.make_result_record(r["gt_class_ids"], r["gt_RTs"], r["gt_scales"], None, r["gt_class_ids"],
                                          r["gt_RTs"], r["gt_scales"], r["gt_handle_visibility"]) for r in g["results"]]
    assert np.allclose(metrics.pose_mAP(perfect, g["synset_names"])[1:, 0, 0], 1.0)


def test_oriented_box_iou_matches_the_reference_toolkit():
    """box_iou_3d / iou_3d against compute_3d_iou_new (utils/util.py:475-547 -> utils/iou.py, utils/box.py) on random and
    special box pairs, every class symmetry rule (golden: tests/golden/make_golden_map.py)."""
    import pickle
    from cppf2_amd import metrics
    with open(os.path.join(GOLDEN, "map_results.pkl"), "rb") as f:
        g = pickle.load(f)
    worst = 0.0
    for p in g["iou_pairs"]:
        got = metrics.iou_3d(p["RT1"], p["RT2"], p["s1"], p["s2"], p["hv"], p["cls"], p["cls"])
        worst = max(worst, abs(got - p["iou"]))
    assert worst < 1e-9, worst
    tail = [p["iou"] for p in g["iou_pairs"][-5:]]
    assert abs(tail[0] - 1.0) < 1e-12 and tail[1] == 0.0 and abs(tail[2] - 0.06) < 1e-12 and abs(tail[3] - 1 / 3) < 1e-12
    # symmetric class: a rotation about y that is a multiple of 10 degrees does not change the score
    p = g["iou_pairs"][0]
    assert p["cls"] == "bottle"
    turned = p["RT1"] @ metrics._y_rotation(np.radians(50.0))
    assert abs(metrics.iou_3d(turned, p["RT2"], p["s1"], p["s2"], 1, "bottle", "bottle") - p["iou"]) < 1e-9


def test_degree_cm_map_with_iou_matches_the_reference_toolkit():
    """compute_degree_cm_mAP's two outputs (3-D IoU AP at 101 thresholds, pose AP) for use_matches_for_pose False and
    True (the latter is what eval.py:400-411 calls) on 40 synthetic images."""
    import pickle
    from cppf2_amd import metrics
    with open(os.path.join(GOLDEN, "map_results.pkl"), "rb") as f:
        g = pickle.load(f)
    iou_aps, pose_aps = metrics.degree_cm_mAP(g["results"], g["synset_names"], (5, 10, 15), (5, 10, 15),
                                              g["iou_thresholds"], 0.1, False)
    assert np.allclose(iou_aps, g["iou_aps"], rtol=0, atol=1e-9, equal_nan=True)
    assert np.allclose(pose_aps, g["pose_aps"], rtol=0, atol=1e-12, equal_nan=True)
    assert np.allclose(pose_aps, metrics.pose_mAP(g["results"], g["synset_names"]), equal_nan=True)
    iou_m, pose_m = metrics.degree_cm_mAP(g["results"], g["synset_names"], (5, 10, 15), (5, 10, 15), g["iou_thresholds"],
                                          0.1, True)
    assert np.allclose(iou_m, g["iou_aps_matched"], rtol=0, atol=1e-9, equal_nan=True)
    assert np.allclose(pose_m, g["pose_aps_matched"], rtol=0, atol=1e-12, equal_nan=True)
    assert pose_m[-1, 0, 0] > pose_aps[-1, 0, 0] and 0 < iou_aps[-1, 75] < iou_aps[-1, 50] < iou_aps[-1, 25] <= 1


def test_crop_transform_maps_the_bbox_onto_the_crop():
    """resize_crop's transform (dataset.py:322-337): crop pixel (0, 0) is the corner of the square around the bbox, the crop
    centre the bbox centre."""
    sys.path.insert(0, ROOT)
    import eval as ev
    t = ev.crop_transform((100, 50, 300, 150), padding=0, out_size=256)
    assert np.allclose(t @ [128, 128, 1], [200, 100, 1]) and np.allclose(t @ [0, 0, 1], [100, 0, 1])
    assert np.allclose(np.linalg.inv(t) @ [300, 200, 1], [256, 256, 1])
    t2 = ev.crop_transform((0, 0, 10, 40), padding=0.2, out_size=224)
    assert np.allclose(t2 @ [112, 112, 1], [5, 20, 1]) and np.isclose(t2[0, 0], 48 / 224)


def test_run_dir_layout_and_saved_config(tmp_path):
    """hydra.run.dir = checkpoints/${cat_name} (config/config.yaml:16-22) resolved without hydra, overridable from the command
    line; the resolved cfg is written where eval.py:92,97 reads it."""
    from cppf2_amd.config import load_checkpoint_config, load_config, run_dir, save_run_config
    from cppf2_amd.training import checkpoint_dir
    cfg, hy = load_config(os.path.join(ROOT, "config"), "config", ["category=laptop"], with_hydra=True)
    assert "hydra" not in cfg and run_dir(cfg, hy) == "checkpoints/laptop"
    cfg2, hy2 = load_config(os.path.join(ROOT, "config"), "config", ["category=mug", "hydra.run.dir=ckpts/shot/${cat_name}-num_more-${num_more}"],
                            with_hydra=True)
    assert run_dir(cfg2, hy2) == "ckpts/shot/mug-num_more-3"
    path = save_run_config(cfg2, str(tmp_path / "run"))
    assert path.endswith(os.path.join(".hydra", "config.yaml"))
    back = load_checkpoint_config(path)
    assert back == cfg2 and list(back.right) == [0, 0, 1] and back.res == 2e-3        # mug votes its second axis about z
    assert checkpoint_dir("a/b").replace(os.sep, "/") == "a/b/lightning_logs/version_0/checkpoints"


def test_core_slice_partitions_any_allowed_core_set():
    """bench.py pins rank r of W to the r-th run of the process' allowed cores (cppf2_amd/benchlib/launch.py:core_slice): every
    allowed core goes to exactly one rank, sizes differ by at most one, whatever sched_getaffinity returns -- a non-contiguous
    cpuset, SMT siblings removed, fewer cores than ranks."""
    from cppf2_amd.benchlib.launch import core_slice
    for cores in (range(256), range(8), [3, 5, 6, 7, 40, 41, 42, 43, 44, 100, 101], set(range(0, 128, 2)), [7, 2, 9]):
        cores = list(cores)
        for W in (1, 2, 3, 8):
            parts = [core_slice(cores, r, W) for r in range(W)]
            if len(set(cores)) >= W:
                assert sorted(c for p in parts for c in p) == sorted(set(cores))            # a partition
                sizes = [len(p) for p in parts]
                assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
                assert all(p == sorted(p) for p in parts)
                assert all(a[-1] < b[0] for a, b in zip(parts[:-1], parts[1:]))             # runs of the sorted ids, in rank order
            else:                                                                           # fewer cores than ranks: shared, never empty
                assert all(len(p) == 1 and p[0] in cores for p in parts)
                assert {p[0] for p in parts} == set(cores)
    assert core_slice([], 0, 8) == [] and core_slice(range(8), 8, 8) == [] and core_slice(range(8), -1, 8) == []
    # the usual 8-GPU host: 2 sockets x 64 cores, socket-major ids -> ranks 0-3 on socket 0, 4-7 on socket 1
    assert core_slice(range(128), 3, 8) == list(range(48, 64)) and core_slice(range(128), 4, 8) == list(range(64, 80))


def test_pin_rank_to_cores_reports_what_it_set(monkeypatch):
    import os
    from cppf2_amd.benchlib import launch
    allowed = {1, 2, 3, 10, 11, 12, 13}
    seen = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(allowed))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: seen.update(cores=list(cores)))
    monkeypatch.delenv("CPPF_BENCH_NO_AFFINITY", raising=False)
    got = launch.pin_rank_to_cores(1, 2)
    assert seen["cores"] == [10, 11, 12, 13] and got == {"cores": 4, "first": 10, "last": 13, "contiguous": True}
    got = launch.pin_rank_to_cores(0, 2)
    assert seen["cores"] == [1, 2, 3] and got["contiguous"]
    assert launch.pin_rank_to_cores(0, 1) is None                                           # one rank: affinity left alone
    monkeypatch.setenv("CPPF_BENCH_NO_AFFINITY", "1")
    assert launch.pin_rank_to_cores(1, 2) is None


def test_voxel_density_scene_generator():
    """cppf2_amd.synth.make_scene_voxel2mm (bench.py --cloud voxel2mm / value_voxel_density): seeded, the same keys as make_scene,
    and the neighbour density eval.py:185-201's 2 mm voxel grid gives real inputs (docs/measurements.md 3: ~250 inside the 2 cm
    SHOT support, against ~90 for the bench's uniform surface samples)."""
    from scipy.spatial import cKDTree
    from cppf2_amd import synth
    a, b = synth.make_scene_voxel2mm(0, 3, 4096), synth.make_scene_voxel2mm(0, 3, 4096)
    assert set(a) == set(synth.make_scene(0, 3, 64)) and a["pc"].dtype == np.float32 and a["pc"].shape == (4096, 3)
    assert np.array_equal(a["pc"], b["pc"]) and not np.array_equal(a["pc"], synth.make_scene_voxel2mm(0, 4, 4096)["pc"])
    nn = np.array([len(x) for x in cKDTree(a["pc"]).query_ball_point(a["pc"], 0.02)])
    assert 230 < nn.mean() < 280 and nn.max() <= 400                          # lists of the long-list kernels' capacity class
    ns = np.array([len(x) for x in cKDTree(synth.make_scene(0, 3, 4096)["pc"]).query_ball_point(synth.make_scene(0, 3, 4096)["pc"], 0.02)])
    assert 70 < ns.mean() < 110
    # points sit on the 2 mm lattice of the camera frame up to the +-0.2 mm jitter; canonical coordinates map back through (R, t, diag)
    off = np.abs(a["pc"] / 2e-3 - np.round(a["pc"] / 2e-3))
    assert off.max() <= 0.1001 + 1e-3
    back = (a["pc"].astype(np.float64) - a["t"]) @ a["R"] / a["diag"]
    assert np.allclose(back, a["pc_canon"], atol=1e-6) and np.abs(a["pc_canon"]).max() <= 0.5


def test_bench_command_line_defaults():
    """The driver's contract: `python bench.py` with no flags = N = 1, a K / W that finish in minutes, the headline workload on the
    synthetic clouds with the voxel-density figure beside it; every round-5 switch exists."""
    from cppf2_amd.benchlib import launch
    a = launch.parse([])
    assert (a.gpus, a.steps, a.warmup, a.scenes_per_gpu, a.points, a.tuples, a.rots) == (1, 100, 3, 64, 4096, 20000, 180)
    assert a.workload == "shot" and a.cloud == "synthetic" and not a.no_voxel_density and not a.separate_encode and not a.single_stream
    assert a.mlp_reserve_cus is None and launch.parse(["--mlp-reserve-cus", "0"]).mlp_reserve_cus == 0   # None: one per shader engine
    b = launch.parse(["--gpus", "8", "--cloud", "voxel2mm", "--separate-encode", "--no-voxel-density", "--workload", "ensemble"])
    assert b.gpus == 8 and b.cloud == "voxel2mm" and b.separate_encode and b.no_voxel_density and b.workload == "ensemble"


def test_bin_prior_dense_is_the_gaussian_logit_bump():
    """ops.BinPrior.dense(): -0.5 ((k - pos) * inv_sigma)^2 in float32, these operations in this order (what the fused bin draw
    generates in its epilogue); equal to synth.teacher_logits' array up to the rounding of its division."""
    import torch
    from cppf2_amd import ops, synth
    rng = np.random.default_rng(0)
    pos = (rng.random((50, 6)) * 36 - 2).astype(np.float32)
    prior = ops.BinPrior(torch.from_numpy(pos), 1.0 / 0.6)
    d = prior.dense(32).numpy()
    inv = np.float32(1.0 / 0.6)
    k = np.arange(32, dtype=np.float32)
    z = (k[None, None, :] - pos[..., None]) * inv
    assert d.dtype == np.float32 and d.shape == (50, 6, 32) and np.array_equal(d, (z * z) * np.float32(-0.5))
    ref = -0.5 * ((k[None, None, :] - pos[..., None]) / np.float32(0.6)) ** 2
    assert np.abs(d - ref).max() <= 4e-6 * np.abs(ref).max()
    try:
        ops.BinPrior(torch.zeros((4, 5)), 1.0)
    except AssertionError:
        pass
    else:
        raise AssertionError("a [T, 5] position array must be refused")
