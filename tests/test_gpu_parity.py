"""HIP path (through the C ABI, via cppf2_amd.ops / cppf2_amd.pipeline) against the oracle and the golden
vectors of the real reference.  Needs an MI355X: run with `pytest -m gpu`.

Bars: bit-exact for integer / index work and for float32 outputs of exact IEEE operation chains; stated
tolerances where libm functions (exp, tan, acos) are involved.
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():          # collected but skipped on CPU boxes (they run -m "not gpu" anyway)
    pytest.skip("no HIP device", allow_module_level=True)

from oracle import cppf_oracle as O          # noqa: E402  (checker only)
from cppf2_amd import ops, synth             # noqa: E402
from cppf2_amd.pipeline import VotingPipeline  # noqa: E402

UP, RIGHT, FRONT = [0, 1, 0], [1, 0, 0], [0, 0, 1]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = torch.device("cuda")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dev(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV, dtype)


# ------------------------------------------------------------------------------------------ a1
@pytest.mark.parametrize("n_points,T,k", [(4096, 20000, 5), (257, 1000, 2), (50000, 333, 8)])
def test_sample_tuples_bit_exact(n_points, T, k):
    got = ops.sample_tuples(n_points, T, k, seed=0x1234567890ABCDEF, scene_ids=(3, 5, 7)).cpu().numpy()
    for i, sid in enumerate((3, 5, 7)):
        want = O.sample_tuples(0x1234567890ABCDEF, sid, T, k, n_points)
        assert np.array_equal(got[i * T:(i + 1) * T], want)
    assert got.min() >= 0 and got.max() < n_points


def test_philox_uniform_bit_exact():
    got = ops.philox_uniform(1000, 6, seed=42, stream_id=7, scene_ids=(0, 1)).cpu().numpy()
    for sid in (0, 1):
        assert np.array_equal(got[sid * 1000:(sid + 1) * 1000], O.philox_uniform(42, sid, 7, 1000, 6))
    assert got.min() >= 0.0 and got.max() < 1.0


# ------------------------------------------------------------------------------------------ a3
def test_encode_shot_golden_bit_exact(small):
    enc = ops.encode_tuples_shot(small["small_pc"], small["small_idx"], small["small_feat"], small["small_normal"])
    assert np.array_equal(enc.cpu().numpy(), small["small_encode_shot"])


def test_encode_coord_bit_exact(small):
    enc = ops.encode_tuples_coord(small["small_pc"], small["small_idx"]).cpu().numpy()
    assert np.array_equal(enc, small["small_encode_shot"][:, :30])
    # into the leading columns of a wider row (DINO layout, 30 + 256)
    out = torch.full((512, 286), 7.0, device=DEV)
    ops.encode_tuples_coord(small["small_pc"], small["small_idx"], out=out)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, :30], small["small_encode_shot"][:, :30]) and np.all(o[:, 30:] == 7.0)


@pytest.mark.parametrize("k,F", [(2, 4), (3, 64), (8, 16)])
def test_encode_shot_other_shapes(k, F):
    rng = np.random.RandomState(k)
    pc = rng.rand(100, 3).astype(np.float32)
    nrm = rng.randn(100, 3).astype(np.float32)
    feat = rng.randn(100, F).astype(np.float32)
    idx = rng.randint(0, 100, (77, k))
    got = ops.encode_tuples_shot(pc, idx, feat, nrm).cpu().numpy()
    assert np.array_equal(got, O.prepare_tuple_inputs_shot(pc, idx, feat, nrm))


# ------------------------------------------------------------------------------------------ a5
def test_generate_target_pairs_golden(small):
    tr, rot = ops.generate_target_pairs(small["small_scaled"], UP, FRONT, RIGHT)
    assert tr.dtype == np.float32 and rot.dtype == np.float32
    assert np.array_equal(tr, small["small_tr0"])                      # +,-,*,/,sqrt in f64: exact
    # acos in f64 rounded to f32: device libm may differ in the last f32 ulp
    assert np.allclose(rot, small["small_rot0"], rtol=0, atol=2.4e-7, equal_nan=True)
    pairs = small["small_pc"][small["small_idx"][:, :2]]
    tr, rot = ops.generate_target_pairs(pairs, UP, FRONT, RIGHT, small["small_centre"])
    assert np.array_equal(tr, small["small_tr1"], equal_nan=True)
    assert np.allclose(rot, small["small_rot1"], rtol=0, atol=2.4e-7, equal_nan=True)


# ------------------------------------------------------------------------------------------ a4
def test_decode_bins_vs_oracle():
    rng = np.random.RandomState(5)
    scene = synth.make_scene(1, 2, 512)
    pc = scene["pc"]
    T = 4000
    idx = rng.randint(0, 512, (T, 5))
    idx[7, 1] = idx[7, 0]
    logits = (rng.randn(T, 6, 32) * 3).astype(np.float32)
    u = rng.rand(T, 6).astype(np.float32)
    u[0] = 0.0
    u[1] = np.float32(1.0 - 2.0 ** -24)
    got = ops.decode_bins(logits, u, pc, idx, UP, FRONT, RIGHT)
    bins, pred, scale, scaled, margin = O.decode_bins(logits, u, pc[idx[:, :2]], return_margin=True)
    gb = got["bins"].cpu().numpy()
    diff = gb != bins
    # expf differs by ulps between libm's: a draw may flip only where the uniform sits on a CDF edge
    assert np.all(margin[diff] < 1e-5) and diff.mean() < 1e-3
    same = ~diff.any(1)
    tr, rot = O.generate_target_pairs(scaled, UP, FRONT, RIGHT)
    assert np.array_equal(got["scale"].cpu().numpy()[same], scale[same])
    assert np.array_equal(got["pred_pairs_scaled"].cpu().numpy()[same], scaled[same])
    assert np.array_equal(got["targets_tr"].cpu().numpy()[same], tr[same], equal_nan=True)
    assert np.allclose(got["targets_rot"].cpu().numpy()[same], rot[same], rtol=0, atol=2.4e-7, equal_nan=True)


def test_decode_bins_generic_nb():
    rng = np.random.RandomState(6)
    pc = rng.rand(64, 3).astype(np.float32)
    idx = rng.randint(0, 64, (300, 5))
    logits = rng.randn(300, 6, 20).astype(np.float32)
    u = rng.rand(300, 6).astype(np.float32)
    got = ops.decode_bins(logits, u, pc, idx, UP, FRONT, RIGHT)["bins"].cpu().numpy()
    bins, _, _, _, margin = O.decode_bins(logits, u, pc[idx[:, :2]], return_margin=True)
    assert np.all(margin[got != bins] < 1e-5)


# ------------------------------------------------------------------------------------------ a6
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_vote_center_golden_small(small, mode):
    grid, cand = ops.vote_center(small["small_pc"], small["small_tr0"], 2e-3, small["small_idx"][:, :2], 36,
                                 trig=(small["small_cos"], small["small_sin"]), mode=mode)
    assert grid.dtype == np.int64 and cand.dtype == np.float64
    assert np.array_equal(grid, small["small_grid_obj"])
    assert np.array_equal(cand, small["small_T_est"])


def _full_inputs(f):
    scene = synth.make_scene(f["seed"], f["scene"], n_points=f["N"])
    pc = scene["pc"]
    idx = synth.host_sample_tuples(f["seed"], f["scene"], f["T"], 5, f["N"]).astype(np.int64)
    scaled = np.load(os.path.join(GOLDEN, "full_scaled.npz"))["scaled"]
    g = np.load(os.path.join(GOLDEN, "small.npz"))
    return scene, pc, idx, scaled, (g["cos180"], g["sin180"]), g["sphere_pts"]


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_vote_center_full_size_golden(full_summary, mode):
    f = full_summary["full"]
    scene, pc, idx, scaled, trig, _ = _full_inputs(f)
    assert sha(pc) == f["pc_sha"] and sha(idx) == f["idx_sha"]
    # device sampler reproduces the table too
    assert np.array_equal(ops.sample_tuples(f["N"], f["T"], 5, f["seed"], (f["scene"],)).cpu().numpy(), idx)
    tr, rot = ops.generate_target_pairs(scaled, UP, FRONT, RIGHT)
    assert sha(tr) == f["targets_tr_sha"]
    grid, T_est = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=mode)
    assert list(grid.shape) == f["grid_shape"]
    assert int(grid.sum()) == f["grid_total"] and int(grid.max()) == f["grid_max"]
    assert int(np.argmax(grid)) == f["grid_argmax"]
    assert sha(grid.astype(np.int64)) == f["grid_sha"]
    assert T_est.tolist() == f["T_est"]


def test_vote_center_default_table_matches_oracle_with_same_table():
    """Without an injected table the product builds train_dino.py:194-195's table with torch on the host;
    the oracle run with that very table must give the identical grid."""
    rng = np.random.RandomState(11)
    scene = synth.make_scene(3, 1, 700)
    pc = scene["pc"]
    idx = rng.randint(0, 700, (3000, 2))
    tr, _ = O.generate_target_pairs(scene["pc_canon"][idx] * np.float32(scene["diag"]), UP, FRONT, RIGHT)
    cs, sn = ops.rotation_table(72)
    grid, cand = ops.vote_center(pc, tr, 3e-3, idx, 72)
    g2, c2 = O.vote_center(pc, tr, 3e-3, idx, 72, trig=(cs.cpu().numpy(), sn.cpu().numpy()))
    assert np.array_equal(grid, g2) and np.array_equal(cand, c2)


def test_vote_center_edge_cases():
    # all pairs degenerate (a == b) or radius <= res: empty grid, argmax 0, world = min corner
    pc = np.random.RandomState(0).rand(50, 3).astype(np.float32) * 0.05
    idx = np.stack([np.arange(50), np.arange(50)], -1)
    tr = np.ones((50, 2), np.float32)
    grid, cand = ops.vote_center(pc, tr, 2e-3, idx, 36)
    assert grid.sum() == 0 and np.array_equal(cand, pc.min(0).astype(np.float64))
    idx = np.stack([np.arange(50), (np.arange(50) + 1) % 50], -1)
    tr = np.full((50, 2), 1e-3, np.float32)        # odist <= res
    g1, c1 = ops.vote_center(pc, tr, 2e-3, idx, 36)
    g2, c2 = O.vote_center(pc, tr, 2e-3, idx, 36, trig=[t.cpu().numpy() for t in ops.rotation_table(36)])
    assert g1.sum() == 0 and np.array_equal(g1, g2) and np.array_equal(c1, c2)
    # NaN / inf vote parameters vote nowhere
    tr = np.full((50, 2), np.nan, np.float32)
    tr[::2] = np.inf
    g1, _ = ops.vote_center(pc, tr, 2e-3, idx, 36)
    assert g1.sum() == 0


# ------------------------------------------------------------------------------------------ a8/a9 wrappers
@pytest.mark.parametrize("name,col", [("up", 0), ("right", 2)])
def test_vote_rotation_and_get_topk_dir_golden(small, name, col):
    mask = small["small_pairs_mask"]
    filt = small["small_idx"][mask]
    rot_f = small["small_rot0"][mask]
    up, vmask = ops.vote_rotation(small["small_pc"], rot_f[:, col], filt[:, :2], 36,
                                  trig=(small["small_cos"], small["small_sin"]))
    assert np.array_equal(vmask.cpu().numpy(), small["small_%s_vmask" % name])
    ref = small["small_%s_cand" % name]
    assert tuple(up.shape) == ref.shape
    assert np.max(np.abs(up.cpu().numpy() - ref)) <= 1e-6          # tanf: a few ulp
    w = np.broadcast_to(small["small_imp_pair_wt"][vmask.cpu().numpy(), None], (ref.shape[0], 36)).reshape(-1, 1)
    # get_topk_dir on the reference's own candidates: identical inputs -> counts bit-exact
    dirs, cnts = ops.get_topk_dir(ref.reshape(-1, 3), small["sphere_pts"], 100000, 1.0, w, topk=5)
    assert np.array_equal(cnts, small["small_%s_top5_counts" % name])
    assert np.array_equal(dirs, small["small_%s_top5_dirs" % name])
    allc = ops.sphere_counts(ref.reshape(-1, 3), small["sphere_pts"], 100000, 1.0, w).cpu().numpy()
    assert np.array_equal(allc, small["small_%s_counts" % name])
    # small chunk size exercises the per-chunk float32 folding
    allc2 = ops.sphere_counts(ref.reshape(-1, 3), small["sphere_pts"], 1000, 1.0, w).cpu().numpy()
    _, _, want2 = O.get_topk_dir(ref.reshape(-1, 3), small["sphere_pts"], 1000, 1.0, w, return_counts=True)
    assert np.array_equal(allc2, want2)
    # no weights
    allc3 = ops.sphere_counts(ref.reshape(-1, 3), small["sphere_pts"], 100000, 1.0).cpu().numpy()
    _, _, want3 = O.get_topk_dir(ref.reshape(-1, 3), small["sphere_pts"], 100000, 1.0, return_counts=True)
    assert np.array_equal(allc3, want3)


def test_vote_rotation_edge_tan_quirk(small):
    up, vm = ops.vote_rotation(small["small_pc"], small["edge_rot"], small["edge_idx"], 36,
                               trig=(small["small_cos"], small["small_sin"]))
    assert np.array_equal(vm.cpu().numpy(), small["edge_vmask"])
    got, ref = up.cpu().numpy(), small["edge_cand"]
    # theta = f32(pi): tan > 0 in the reference (f32 pi rounds above pi) -> votes along +u; check sign + value
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-6
    # theta = f32(pi/2): tan ~ -2.3e7 (huge): direction is well defined, compare loosely
    assert np.max(np.abs(got[1] - ref[1])) <= 1e-4
    assert np.max(np.abs(got[2:] - ref[2:])) <= 1e-6


# ------------------------------------------------------------------------------------------ pipeline
def _scene_inputs(seed, sid, N, T, rng, sigma=0.6):
    scene = synth.make_scene(seed, sid, N)
    idx = synth.host_sample_tuples(seed, sid, T, 5, N)
    logits = synth.teacher_logits(scene["pc_canon"], idx, 32, sigma) + rng.randn(T, 6, 32).astype(np.float32) * 0.3
    u = O.philox_uniform(seed, sid, 1, T, 6)
    scales = rng.rand(T, 3).astype(np.float32)
    return scene, idx, logits.astype(np.float32), u, scales


@pytest.mark.parametrize("use_lut", [False, True])
def test_pipeline_vs_oracle_ragged_batch(use_lut):
    rng = np.random.RandomState(3)
    Ns, Ts = [600, 1024, 333], [3000, 5000, 1111]
    R = 60
    scenes = [_scene_inputs(9, s, Ns[s], Ts[s], rng) for s in range(3)]
    pipe = VotingPipeline(Ns, Ts, k=5, res=2e-3, num_rots=R, cells_cap=1 << 20)
    if not use_lut:
        pipe.lut = None
    pts = dev(np.concatenate([s[0]["pc"] for s in scenes]), torch.float32)
    idx = dev(np.concatenate([s[1] for s in scenes]), torch.int32)
    logits = dev(np.concatenate([s[2] for s in scenes]), torch.float32)
    u = dev(np.concatenate([s[3] for s in scenes]), torch.float32)
    sc = dev(np.concatenate([s[4] for s in scenes]), torch.float32)
    res = pipe.results_to_numpy(pipe.vote(pts, idx, logits, u, sc))
    trig = (pipe.cs.cpu().numpy(), pipe.sn.cpu().numpy())
    bins_all = pipe.bins.cpu().numpy()
    t0 = 0
    for s, (scene, idx_s, lg, us, scl) in enumerate(scenes):
        T = Ts[s]
        # feed the oracle the device's bins (exp() ulps may flip a rare draw; checked in test_decode_*)
        ob = O.decode_bins(lg, us, scene["pc"][idx_s[:, :2]], return_margin=True)
        flips = (ob[0] != bins_all[t0:t0 + T])
        assert np.all(ob[4][flips] < 1e-5)
        onehot = np.full((T, 6, 32), -1e4, np.float32)
        np.put_along_axis(onehot, bins_all[t0:t0 + T, :, None].astype(np.int64), 0.0, -1)
        want = O.run_scene(scene["pc"], idx_s, onehot, scl, us, UP, RIGHT, FRONT, 2e-3, num_rots=R, trig=trig)
        r = res[s]
        assert r["argmax"] == want["argmax"] and r["peak"] == want["grid_obj"].max()
        assert np.array_equal(r["t"], want["T_est"])
        assert r["kept"] == int(want["pairs_mask"].sum())
        assert np.array_equal(pipe.mask.cpu().numpy()[t0:t0 + T].astype(bool), want["pairs_mask"])
        assert np.array_equal(pipe.errs.cpu().numpy()[t0:t0 + T], want["back_errs"])
        assert np.float32(pipe.thr.cpu().numpy()[s]) == np.float32(want["thr"])
        kept = r["kept"]
        assert np.array_equal(pipe.kept_wt.cpu().numpy()[t0:t0 + kept], want["imp_pair_wt"])
        for a, name in ((0, "up"), (1, "right")):
            got_c = pipe.counts[a, s].cpu().numpy()
            d = np.abs(got_c - want[name + "_counts"])
            # tanf ulps flip isolated cone tests: <= 4 bins off by <= 2 votes of the largest weight
            assert (d > 0).sum() <= 4 and d.max() <= 2.0 / want["imp_pair_wt"].min()
            assert int(r[name + "_idx"]) == want[name + "_idx"]
        assert np.allclose(r["R"], want["R_est"], atol=1e-6)
        assert np.array_equal(r["scale"], want["pred_scale"])
        # pose sanity against synthetic ground truth: centre within 5 mm, up axis within 5 deg
        assert np.linalg.norm(r["t"] - scene["t"]) < 5e-3
        cosang = abs(float(r["R"][:, 1] @ scene["R"][:, 1]))
        assert np.degrees(np.arccos(min(cosang, 1.0))) < 5.0
        t0 += T


@pytest.mark.parametrize("cfg_seed", list(range(10)))
def test_pipeline_vs_oracle_random_configurations(cfg_seed):
    """The whole post-MLP path (bin draw, vote parameters, centre vote, arg-max, back-vote filter, both rotation votes, pose) on
    configurations drawn from a seed -- batch size, ragged point / tuple counts down to a few dozen, rotation counts that are not a
    multiple of the vote quantum, cell sizes, axis assignments, noisy teachers -- against the oracle, same bars as the ragged-batch
    test (integers, masks, weights and translations bit for bit; rotation-bin counts up to the tanf cone flips)."""
    rng = np.random.RandomState(1000 + cfg_seed)
    B = int(rng.randint(1, 5))
    Ns = [int(rng.randint(40, 1400)) for _ in range(B)]
    Ts = [int(rng.randint(150, 3600)) for _ in range(B)]
    R = int(rng.choice([5, 8, 13, 36, 59, 97, 120, 181]))
    res = float(rng.choice([1.5e-3, 2e-3, 3e-3, 5e-3]))
    axes = [[0, 1, 0], [1, 0, 0], [0, 0, 1]]
    perm = rng.permutation(3)
    up, right, front = axes[perm[0]], axes[perm[1]], axes[perm[2]]
    sigma = float(rng.choice([0.4, 0.6, 1.5]))
    scenes = [_scene_inputs(50 + cfg_seed, s, Ns[s], Ts[s], rng, sigma=sigma) for s in range(B)]
    pipe = VotingPipeline(Ns, Ts, k=5, res=res, num_rots=R, cells_cap=1 << 21, cfg_up=tuple(up), cfg_right=tuple(right),
                          cfg_front=tuple(front))
    pts = dev(np.concatenate([s[0]["pc"] for s in scenes]), torch.float32)
    idx = dev(np.concatenate([s[1] for s in scenes]), torch.int32)
    logits = dev(np.concatenate([s[2] for s in scenes]), torch.float32)
    u = dev(np.concatenate([s[3] for s in scenes]), torch.float32)
    sc = dev(np.concatenate([s[4] for s in scenes]), torch.float32)
    rec = pipe.results_to_numpy(pipe.vote(pts, idx, logits, u, sc))
    trig = (pipe.cs.cpu().numpy(), pipe.sn.cpu().numpy())
    bins_all = pipe.bins.cpu().numpy()
    t0 = 0
    for s, (scene, idx_s, lg, us, scl) in enumerate(scenes):
        T = Ts[s]
        ob = O.decode_bins(lg, us, scene["pc"][idx_s[:, :2]], return_margin=True)
        flips = (ob[0] != bins_all[t0:t0 + T])
        assert np.all(ob[4][flips] < 1e-5)
        onehot = np.full((T, 6, 32), -1e4, np.float32)
        np.put_along_axis(onehot, bins_all[t0:t0 + T, :, None].astype(np.int64), 0.0, -1)
        want = O.run_scene(scene["pc"], idx_s, onehot, scl, us, up, right, front, res, num_rots=R, trig=trig)
        r = rec[s]
        assert r["argmax"] == want["argmax"] and r["peak"] == want["grid_obj"].max(), (cfg_seed, s)
        assert np.array_equal(r["t"], want["T_est"])
        assert r["kept"] == int(want["pairs_mask"].sum())
        assert np.array_equal(pipe.mask.cpu().numpy()[t0:t0 + T].astype(bool), want["pairs_mask"])
        assert np.array_equal(pipe.errs.cpu().numpy()[t0:t0 + T], want["back_errs"])
        assert np.float32(pipe.thr.cpu().numpy()[s]) == np.float32(want["thr"])
        kept = r["kept"]
        assert np.array_equal(pipe.kept_wt.cpu().numpy()[t0:t0 + kept], want["imp_pair_wt"])
        for a, name in ((0, "up"), (1, "right")):
            got_c = pipe.counts[a, s].cpu().numpy()
            d = np.abs(got_c - want[name + "_counts"])
            assert (d > 0).sum() <= 4 and d.max() <= 2.0 / want["imp_pair_wt"].min(), (cfg_seed, s, name, int((d > 0).sum()), float(d.max()))
            # the arg-max bin may legitimately differ only where the cone flips touch the top count
            if int(r[name + "_idx"]) != want[name + "_idx"]:
                top = np.sort(want[name + "_counts"])[-2:]
                assert top[1] - top[0] <= 2.0 * d.max() + 1e-6, (cfg_seed, s, name)
        assert np.array_equal(r["scale"], want["pred_scale"])
        t0 += T


def test_rot_bins_lut_equals_dense_full_size(full_summary):
    """Size-independent property at the full configuration (4096 x 20k x 180): the windowed search over
    cell->bins lookup table returns exactly the exhaustive counts; and both agree with the golden reference summary."""
    f = full_summary["full"]
    scene, pc, idx, scaled, trig, sphere = _full_inputs(f)
    pipe = VotingPipeline([f["N"]], [f["T"]], res=2e-3, num_rots=f["R"], trig=trig)
    pts, di = dev(pc, torch.float32), dev(idx, torch.int32)
    tr, rot = ops.generate_target_pairs(scaled, UP, FRONT, RIGHT)
    pipe.tr.copy_(dev(tr, torch.float32))
    pipe.rot.copy_(dev(rot, torch.float32))
    pipe.vote_center(pts, di)
    assert int(pipe.argmax.item()) == f["grid_argmax"] and int(pipe.peak.item()) == f["grid_max"]
    pipe.backvote(pts, di)
    assert int(pipe.kept_count.item()) == f["kept"]
    assert sha(pipe.mask.cpu().numpy().astype(bool)) == f["pairs_mask_sha"]
    assert sha(pipe.kept_wt.cpu().numpy()[:f["kept"]]) == f["imp_pair_wt_sha"]
    assert float(pipe.thr.item()) == float(np.float32(f["thr"]))
    assert pipe.lut is not None
    pipe.rot_bins(pts, di, use_lut=False)
    dense = pipe.counts.cpu().numpy().copy()
    pipe.rot_bins(pts, di, use_lut=True)
    win = pipe.counts.cpu().numpy().copy()
    # same hits; float64 partial sums are grouped differently (1e-16 relative) -> identical after the f32 fold
    assert np.array_equal(dense, win)
    pipe.rot_bins(pts, di, use_lut=True)
    assert np.array_equal(win, pipe.counts.cpu().numpy())          # fixed-order fold: run-to-run identical
    fs = np.load(os.path.join(GOLDEN, "full_scaled.npz"))
    wmin = pipe.kept_wt.cpu().numpy()[:f["kept"]].min()
    for a, name in ((0, "up"), (1, "right")):
        d = np.abs(win[a, 0] - fs[name + "_counts"])
        assert (d > 0).sum() <= 4 and d.max() <= 2.0 / wmin
        assert int(pipe.top_idx[a, 0].item()) == f[name + "_top1"]


def test_vote_center_properties_full_size(full_summary):
    """Invariants at full size: the two accumulation strategies agree cell for cell; permuting the tuples
    does not change the grid (integer votes are order-free); total votes = valid votes counted by the oracle."""
    f = full_summary["full"]
    scene, pc, idx, scaled, trig, _ = _full_inputs(f)
    tr, _ = ops.generate_target_pairs(scaled, UP, FRONT, RIGHT)
    g1, c1 = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=1)
    g2, c2 = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=2)
    assert np.array_equal(g1, g2) and np.array_equal(c1, c2)
    perm = np.random.RandomState(0).permutation(f["T"])
    g3, _ = ops.vote_center(pc, tr[perm], 2e-3, idx[perm][:, :2], f["R"], trig=trig, mode=1)
    assert np.array_equal(g1, g3)
    assert int(g1.sum()) == f["grid_total"]


@pytest.mark.parametrize("mode", [1, 2])
def test_weighted_centre_votes_extension(full_summary, mode):
    """BASELINE config 5's uncertainty-weighted accumulator (not in the reference): w == 1 gives exactly 256 x the
    pinned integer grid; random weights match the oracle's fixed-point restatement cell for cell."""
    f = full_summary["full"]
    scene, pc, idx, scaled, trig, _ = _full_inputs(f)
    tr, _ = ops.generate_target_pairs(scaled, UP, FRONT, RIGHT)
    g0, c0 = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=mode)
    g1, c1 = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=mode, weights=np.ones(f["T"], np.float32))
    assert np.array_equal(g1, g0 * 256) and np.array_equal(c0, c1)
    w = np.random.RandomState(1).rand(f["T"]).astype(np.float32) * 2
    w[::7] = 0
    w[3] = 9.0                      # clamped to 4
    gw, cw = ops.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, mode=mode, weights=w)
    go, co = O.vote_center(pc, tr, 2e-3, idx[:, :2], f["R"], trig=trig, weights=w)
    assert np.array_equal(gw, go) and np.array_equal(cw, co)


def test_encode_fp16_feature_table_extension(small):
    """BASELINE config 5 "fp16 features": rows built from a float16 table equal the float32 path on the rounded table."""
    feat16 = torch.from_numpy(small["small_feat"]).half()
    got = ops.encode_tuples_shot(small["small_pc"], small["small_idx"], feat16.cuda(), small["small_normal"]).cpu().numpy()
    want = O.prepare_tuple_inputs_shot(small["small_pc"], small["small_idx"], feat16.float().numpy(), small["small_normal"])
    assert np.array_equal(got, want)


def test_dense_pairs_65536_properties():
    """BASELINE config 5 size (64 k pairs / scene): accumulation strategies agree cell for cell and the lookup-table
    rotation search equals the exhaustive one."""
    rng = np.random.RandomState(2)
    N, T, R = 2048, 65536, 90
    sc, idx, lg, u = (lambda s: (s, synth.host_sample_tuples(11, 0, T, 5, N), None, None))(synth.make_scene(11, 0, N))
    lg = synth.teacher_logits(sc["pc_canon"], idx, 32, 0.6).astype(np.float32)
    u = O.philox_uniform(11, 0, 1, T, 6)
    pipe = VotingPipeline([N], [T], num_rots=R)
    pts, di = dev(sc["pc"], torch.float32), dev(idx, torch.int32)
    pipe.decode(pts, di, dev(lg, torch.float32), dev(u, torch.float32))
    tr = pipe.tr.cpu().numpy()
    g1, c1 = ops.vote_center(sc["pc"], tr, 2e-3, idx[:, :2], R, mode=1)
    g2, c2 = ops.vote_center(sc["pc"], tr, 2e-3, idx[:, :2], R, mode=2)
    assert np.array_equal(g1, g2) and np.array_equal(c1, c2)
    pipe.vote_center(pts, di)
    assert int(pipe.argmax.item()) == int(np.argmax(g1)) and np.linalg.norm(c1 - sc["t"]) < 5e-3
    pipe.backvote(pts, di)
    assert int(pipe.kept_count.item()) in (6553, 6554)
    pipe.rot_bins(pts, di, use_lut=False)
    dense = pipe.counts.cpu().numpy().copy()
    pipe.rot_bins(pts, di, use_lut=True)
    assert np.array_equal(dense, pipe.counts.cpu().numpy())
