"""Edge cases the reference's own code paths imply (empty / tiny / ragged / oversized inputs), on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

from oracle import cppf_oracle as O            # noqa: E402
from oracle import shot_oracle as S            # noqa: E402
from cppf2_amd import ops, shot, synth         # noqa: E402
from cppf2_amd.pipeline import VotingPipeline  # noqa: E402

DEV = torch.device("cuda")


def _inputs(seed, sid, N, T):
    sc = synth.make_scene(seed, sid, N)
    idx = synth.host_sample_tuples(seed, sid, T, 5, N)
    lg = synth.teacher_logits(sc["pc_canon"], idx, 32, 0.6)
    u = O.philox_uniform(seed, sid, 1, T, 6)
    return sc, idx, lg, u


def _run(Ns, Ts, parts, R=36, **kw):
    pipe = VotingPipeline(Ns, Ts, num_rots=R, **kw)
    cat = lambda i, dt, shape: torch.as_tensor(np.concatenate([p[i].reshape(shape) for p in parts]) if parts else
                                               np.zeros(shape, dtype=np.float32)).to(DEV, dt)
    pts = cat(0, torch.float32, (-1, 3))
    idx = cat(1, torch.int32, (-1, 5))
    lg = cat(2, torch.float32, (-1, 6, 32))
    u = cat(3, torch.float32, (-1, 6))
    return pipe, pipe.results_to_numpy(pipe.vote(pts, idx, lg, u))


def test_empty_and_tiny_scenes_inside_a_batch():
    a = _inputs(4, 0, 700, 3000)
    tiny_pc = synth.make_scene(4, 1, 5)["pc"]
    tiny_idx = np.array([[0, 1, 2, 3, 4], [1, 1, 2, 3, 4], [4, 3, 2, 1, 0]], np.int32)
    tiny_lg = np.zeros((3, 6, 32), np.float32)
    tiny_u = np.full((3, 6), 0.5, np.float32)
    z3, z5 = np.zeros((0, 3), np.float32), np.zeros((0, 5), np.int32)
    parts = [(a[0]["pc"], a[1], a[2], a[3]), (z3, z5, np.zeros((0, 6, 32), np.float32), np.zeros((0, 6), np.float32)),
             (tiny_pc, tiny_idx, tiny_lg, tiny_u)]
    pipe, res = _run([700, 0, 5], [3000, 0, 3], parts)
    _, solo = _run([700], [3000], parts[:1])
    for f in ("argmax", "t", "peak", "up_idx", "right_idx", "kept", "R"):
        assert np.array_equal(res[f][0], solo[f][0]), f          # neighbours in the batch do not leak
    assert res["flags"][1] & 1 and res["kept"][1] == 0 and res["peak"][1] == 0
    assert np.all(np.isfinite(res["R"][2])) and res["kept"][2] <= 3


def test_grid_above_cells_cap_is_flagged_not_voted():
    a = _inputs(5, 0, 400, 1000)
    pipe, res = _run([400], [1000], [(a[0]["pc"], a[1], a[2], a[3])], cells_cap=1000)
    assert res["ncell"][0] > 1000 and res["flags"][0] & 4 and res["argmax"][0] == 0


def test_huge_grid_takes_the_global_atomic_path_and_matches_oracle():
    # res = 0.5 mm -> ~2e7 cells: more than 64 LDS slabs, mode 0 selects global atomics
    sc, idx, lg, u = _inputs(6, 0, 300, 800)
    tr, _ = O.generate_target_pairs(sc["pc_canon"][idx[:, :2]] * np.float32(sc["diag"]), [0, 1, 0], [0, 0, 1], [1, 0, 0])
    cs, sn = ops.rotation_table(24)
    grid, cand = ops.vote_center(sc["pc"], tr, 5e-4, idx[:, :2], 24)
    g2, c2 = O.vote_center(sc["pc"], tr, 5e-4, idx[:, :2], 24, trig=(cs.cpu().numpy(), sn.cpu().numpy()))
    assert grid.size > 64 * 36864 and np.array_equal(grid, g2) and np.array_equal(cand, c2)


def _shot_oracle(pc, rn, rs, arithmetic):
    """(descriptors, normals, diagnostics) of the oracle in the arithmetic the kernel was asked for."""
    s_, n_, _, d_ = S.compute_ex(pc, rn, rs, pcl_arithmetic=(arithmetic == "pcl"))
    return s_, n_, d_


# normals: float64 Jacobi path to float rounding; PCL arithmetic = identical float32 sums, device atan2f / cosf / sinf inside the roots
NORMAL_TOL = {"f64": 2e-6, "pcl": 2e-5}


def _desc_close(hs, os_, d, arithmetic):
    """Descriptor rows within 2e-5, except (PCL arithmetic) rows where a neighbour's normal -- equal to 1e-5 only -- sits within that
    distance of one of PCL's cosine steps (the oracle's own boundary margin)."""
    err = np.abs(hs - os_).max(1)
    if arithmetic == "f64":
        return bool(np.all(err < 2e-5))
    exempt = (d[:, 5] < 2e-5) | (d[:, 8] < 4e-7)
    return bool(np.all(err[~exempt] < 5e-5)) and (err >= 5e-5).mean() < 2e-2


@pytest.mark.parametrize("arithmetic", ["f64", "pcl"])
def test_shot_large_sparse_cloud_coarsened_cells(arithmetic):
    # a cloud spanning far more than CELL_CAP cells of edge r: the cell edge is coarsened, results must not change
    rng = np.random.RandomState(0)
    pc = (rng.rand(6000, 3) * np.float32([1.0, 1.5, 1.2])).astype(np.float32)
    pc[:3000] = pc[:3000] * 0.05 + 0.4                     # one dense blob so that descriptors exist
    hs, hn = shot.compute(pc, 0.02, 0.02, arithmetic=arithmetic)
    os_, on, d = _shot_oracle(pc, 0.02, 0.02, arithmetic)
    hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
    assert np.array_equal(np.isnan(os_), np.isnan(hs)) and np.array_equal(np.isnan(on), np.isnan(hn))
    ok = ~np.isnan(os_).any(1)
    assert ok.sum() > 1000
    assert _desc_close(hs[ok], os_[ok], d[ok], arithmetic)
    assert np.allclose(hn, on, atol=NORMAL_TOL[arithmetic], equal_nan=True)


@pytest.mark.parametrize("arithmetic", ["f64", "pcl"])
def test_shot_different_radii(arithmetic):
    sc = synth.make_scene(8, 0, 900)
    hs, hn = shot.compute(sc["pc"], 0.012, 0.025, arithmetic=arithmetic)
    os_, on, d = _shot_oracle(sc["pc"], 0.012, 0.025, arithmetic)
    ok = ~np.isnan(os_).any(1)
    assert np.allclose(hn.reshape(-1, 3), on, atol=NORMAL_TOL[arithmetic], equal_nan=True)
    assert _desc_close(hs.reshape(-1, 352)[ok], os_[ok], d[ok], arithmetic)


@pytest.mark.parametrize("arithmetic", ["f64", "pcl"])
def test_shot_tie_break_with_many_equal_distances(arithmetic):
    """The sign tie-break of the local reference frame (the 5 neighbours around the median of the (distance, index) order) where
    shot_hist's selection by distance buckets cannot narrow the candidates down: 220 neighbours in antipodal pairs on a ring around
    the query (sign counts tie exactly; all distances fall into one or two of the 64 buckets, more than the 64 keys the selection
    holds), so the exhaustive count decides -- against the oracle, like every other row of the cloud."""
    q = 2.0 ** -14
    rng = np.random.RandomState(3)
    n_pairs = 110
    th = rng.rand(n_pairs) * np.pi
    off = np.stack([np.cos(th) * 0.01, np.sin(th) * 0.01, (rng.rand(n_pairs) - 0.5) * 0.004], 1)
    off = (np.round(off / q) * q).astype(np.float32)
    c = np.float32([0.5, 0.5, 0.5])
    pc = np.concatenate([c[None], c + off, c - off]).astype(np.float32)
    assert np.array_equal(pc[1:1 + n_pairs] - c, -(pc[1 + n_pairs:] - c))          # exact antipodal pairs in float32
    d2 = ((pc[1:] - c) ** 2).sum(1)
    assert d2.max() - d2.min() < 2 * 0.02 ** 2 / 64                                  # at most two distance buckets
    hs, hn = shot.compute(pc, 0.02, 0.02, arithmetic=arithmetic)
    os_, on, d = _shot_oracle(pc, 0.02, 0.02, arithmetic)
    hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
    assert np.array_equal(np.isnan(os_), np.isnan(hs)) and not np.isnan(os_[0]).any()
    assert np.allclose(hn, on, atol=NORMAL_TOL[arithmetic], equal_nan=True)
    ok = ~np.isnan(os_).any(1)
    assert _desc_close(hs[ok], os_[ok], d[ok], arithmetic)
    assert np.abs(hs[0] - os_[0]).max() < 5e-5                                       # the ring's centre: the tied query


def _random_cloud(rng):
    """A cloud drawn from a seed: shape (ball, shell, noisy plane, two crossing planes, cylinder), size, density and radii."""
    n = int(rng.randint(60, 3500))
    kind = int(rng.randint(5))
    ext = float(rng.choice([0.03, 0.06, 0.12, 0.25]))
    if kind == 0:
        v = rng.randn(n, 3)
        pc = v / np.linalg.norm(v, axis=1, keepdims=True) * (rng.rand(n, 1) ** (1 / 3)) * ext
    elif kind == 1:
        v = rng.randn(n, 3)
        pc = v / np.linalg.norm(v, axis=1, keepdims=True) * ext + rng.randn(n, 3) * 2e-4
    elif kind == 2:
        pc = np.concatenate([(rng.rand(n, 2) - 0.5) * 2 * ext, rng.randn(n, 1) * 3e-4], 1)
    elif kind == 3:
        a = np.concatenate([(rng.rand(n // 2, 2) - 0.5) * 2 * ext, rng.randn(n // 2, 1) * 2e-4], 1)
        b = np.concatenate([rng.randn(n - n // 2, 1) * 2e-4, (rng.rand(n - n // 2, 2) - 0.5) * 2 * ext], 1)
        pc = np.concatenate([a, b])
    else:
        th, z = rng.rand(n) * 2 * np.pi, (rng.rand(n) - 0.5) * 3 * ext
        pc = np.stack([np.cos(th) * ext * 0.4, z, np.sin(th) * ext * 0.4], 1) + rng.randn(n, 3) * 2e-4
    q, _ = np.linalg.qr(rng.randn(3, 3))
    pc = pc @ q.T + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.4, 1.5)])
    rn = float(rng.choice([0.008, 0.012, 0.02, 0.03]))
    rs = float(rng.choice([0.01, 0.02, 0.035]))
    pc = pc.astype(np.float32)
    # thinned until no neighbour list (inside the larger radius) exceeds the 512 entries the PCL-arithmetic path ranks: longer
    # lists keep the float64 sums by design (test_shot_neighbour_list_paths covers them)
    rm2 = np.float32(max(rn, rs)) ** 2
    while True:
        cnt = (((pc[:, None, :] - pc[None, :, :]) ** 2).sum(-1) < rm2).sum(1)
        if cnt.max() <= 500:
            break
        pc = pc[rng.rand(pc.shape[0]) < 0.7]
    return pc, rn, rs


@pytest.mark.parametrize("arithmetic", ["f64", "pcl"])
@pytest.mark.parametrize("cfg_seed", list(range(6)))
def test_shot_random_clouds_vs_oracle(cfg_seed, arithmetic):
    """shot.compute on clouds drawn from a seed (shape, size 60-3500 points, extent 3-25 cm, both radii) against the oracle in the same
    arithmetic: NaN pattern equal, normals to the arithmetic's tolerance, descriptor rows to 2e-5 (PCL arithmetic: 5e-5 with the
    oracle's own boundary margin)."""
    pc, rn, rs = _random_cloud(np.random.RandomState(700 + cfg_seed))
    hs, hn = shot.compute(pc, rn, rs, arithmetic=arithmetic)
    os_, on, d = _shot_oracle(pc, rn, rs, arithmetic)
    hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
    assert np.array_equal(np.isnan(os_), np.isnan(hs)) and np.array_equal(np.isnan(on), np.isnan(hn))
    assert np.allclose(hn, on, atol=NORMAL_TOL[arithmetic], equal_nan=True), float(np.nanmax(np.abs(hn - on)))
    ok = ~np.isnan(os_).any(1)
    if ok.any() and arithmetic == "pcl":
        assert _desc_close(hs[ok], os_[ok], d[ok], arithmetic)
    elif ok.any():
        # float64 normals: rows within 2e-5 except where the oracle itself has a neighbour on a decision boundary of PCL's
        # interpolation (its margins, as in tests/test_shot.py::test_hip_shot_vs_oracle)
        err = np.abs(hs[ok] - os_[ok]).max(1)
        exempt = (d[ok, 5] < 1e-6) | (d[ok, 8] < 4e-7)
        assert np.all(err[~exempt] < 2e-5), float(err[~exempt].max())
        assert (err >= 2e-5).mean() < 5e-3


def test_vote_center_two_call_form_equals_single_call():
    """CPPF_VC_FRAMES_ONLY + CPPF_VC_FRAMES_READY (the form bench.py times the vote kernel with) == one call."""
    from cppf2_amd import ops, synth
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    B, N, T = 3, 1500, 6000
    scs = [synth.make_scene(4, b, N) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = ops.sample_tuples(N, T, 5, 4, tuple(range(B)))
    lg = torch.cat([torch.from_numpy(synth.teacher_logits(s["pc_canon"], idx[b * T:(b + 1) * T].cpu().numpy(), 32))
                    for b, s in enumerate(scs)]).to(dev)
    u = ops.philox_uniform(T, 6, 4, 1, tuple(range(B)))
    pipe = VotingPipeline([N] * B, [T] * B, num_rots=60)
    pipe.decode(pts, idx, lg, u)
    pipe.vote_center(pts, idx)
    want = (pipe.argmax.clone(), pipe.peak.clone(), pipe.world.clone())
    pipe.argmax.zero_(); pipe.peak.zero_(); pipe.world.zero_()
    pipe.vote_center(pts, idx, phase=1)
    pipe.vote_center(pts, idx, phase=2)
    assert torch.equal(pipe.argmax, want[0]) and torch.equal(pipe.peak, want[1]) and torch.equal(pipe.world, want[2])
    assert int(pipe.peak.min()) > 0


@pytest.mark.parametrize("nb", [32, 20])
def test_decode_logit_prior_equals_adding_it_first(nb):
    """cppf_decode_bins(logits, logit_prior) == cppf_decode_bins(logits + logit_prior): one float32 add per logit."""
    rng = np.random.RandomState(8)
    N, T = 300, 2000
    pc = torch.from_numpy(rng.rand(N, 3).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.randint(0, N, (T, 5)).astype(np.int32)).cuda()
    lg = torch.from_numpy(rng.randn(T, 6, nb).astype(np.float32) * 3).cuda()
    pr = torch.from_numpy(rng.randn(T, 6, nb).astype(np.float32) * 3).cuda()
    u = torch.from_numpy(rng.rand(T, 6).astype(np.float32)).cuda()
    a = ops.decode_bins(lg, u, pc, idx, (0, 1, 0), (0, 0, 1), (1, 0, 0), prior=pr)
    b = ops.decode_bins(lg + pr, u, pc, idx, (0, 1, 0), (0, 0, 1), (1, 0, 0))
    for k_ in a:
        assert torch.equal(a[k_], b[k_]), k_


def test_shot_describe_nan_to_zero_equals_nan_to_num():
    """describe(nan_to_zero=True) == nan_to_num(describe()) (eval.py:215), on a cloud with isolated points (NaN rows)."""
    rng = np.random.RandomState(5)
    dense = rng.rand(600, 3).astype(np.float32) * 0.05
    lonely = (rng.rand(12, 3).astype(np.float32) + 3.0) * np.arange(1, 13, dtype=np.float32)[:, None]
    pts = torch.from_numpy(np.concatenate([dense, lonely])).cuda()
    off = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device="cuda")
    nrm = shot.prepare_device(pts, off, 0.02, 0.02)
    a = shot.describe_device(pts, off, nrm, 0.02)
    assert torch.isnan(a).any()
    shot.prepare_device(pts, off, 0.02, 0.02)
    b = shot.describe_device(pts, off, nrm, 0.02, nan_to_zero=True)
    assert not torch.isnan(b).any()
    assert torch.equal(torch.nan_to_num(a, nan=0.0), b)


@pytest.mark.parametrize("n, rn, rs", [(2600, 0.02, 0.02),      # 512 < neighbours <= 1024: the shared list overflows, hist rebuilds its own
                                       (6000, 0.02, 0.02),      # > 1024: LDS list overflows too, both kernels rescan the runs
                                       (1500, 0.03, 0.012)])    # normal radius > descriptor radius: hist filters cov's list
def test_shot_neighbour_list_paths(n, rn, rs):
    rng = np.random.RandomState(n)
    v = rng.randn(n, 3)
    pc = (v / np.linalg.norm(v, axis=1, keepdims=True) * (rng.rand(n, 1) ** (1 / 3)) * 0.03 + 0.5).astype(np.float32)
    hs, hn = shot.compute(pc, rn, rs, arithmetic="f64")
    os_, on, _ = S.compute(pc, rn, rs)
    hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
    assert np.array_equal(np.isnan(os_), np.isnan(hs))
    ok = ~np.isnan(os_).any(1)
    assert ok.sum() > n // 2
    # the normals in PCL's arithmetic on the same clouds: lists up to 313 neighbours are ranked in LDS, up to 512 read back from
    # the workspace copy, longer ones (the 2600- and 6000-point balls) keep the float64 sums -- documented, and visible here as
    # agreement with the float64 oracle where the PCL-arithmetic oracle is ~1e-4 away
    pn = shot.estimate_normal(pc, rn, arithmetic="pcl").reshape(-1, 3)
    _, on1, _, _ = S.compute_ex(pc, rn, rs, pcl_arithmetic=True)
    cnt_n = ((pc[:, None, :] - pc[None, :, :]) ** 2).sum(-1) < np.float32(rn) * np.float32(rn)
    short = cnt_n.sum(1) <= 512
    if short.any():
        assert np.allclose(pn[short], on1[short], atol=2e-5, equal_nan=True)
    if (~short).any():
        assert np.allclose(pn[~short], on[~short], atol=2e-6, equal_nan=True)
    d2 = ((pc[:, None, :] - pc[None, :200, :]) ** 2).sum(-1)
    cnt = (d2 < rs * rs).sum(0)
    if n == 2600:
        assert 512 < cnt.max() <= 1024
    if n == 6000:
        assert cnt.max() > 1024
    # PCL's SHOT is a discontinuous function of each neighbour (oracle/shot_oracle.c: shot_accumulate): a neighbour whose
    # cosine / radius / elevation / azimuth lies within float rounding of a decision boundary may land in the adjacent
    # bin (the kernel interpolates in float, the oracle in double like PCL), moving (w - 0.5) / |h| of weight.  Rows may
    # differ ONLY there, or where the LRF itself is ambiguous; every other row agrees to 2e-5.
    _, _, _, diag = S.compute_ex(pc, rn, rs)
    err = np.abs(hs[ok] - os_[ok]).max(1)
    d = diag[ok]
    gap = np.minimum(d[:, 0] - d[:, 1], d[:, 1] - d[:, 2]) / d[:, 0]
    exempt = (d[:, 5] < 1e-6) | (d[:, 8] < 4e-7) | (np.abs(d[:, 3]) <= 1) | (np.abs(d[:, 4]) <= 1) | (gap < 1e-9)
    assert np.all(err[~exempt] < 2e-5), (int((err[~exempt] >= 2e-5).sum()), float(err[~exempt].max()))
    moved = err >= 2e-5
    assert moved.mean() < 5e-3, float(moved.mean())
    # a jump moves at most a few neighbours' total weight (<= 4 each): bounded by 12 / |h|
    assert np.all(err[moved] <= 12.0 / d[moved, 6])
    assert np.allclose(hn[ok], on[ok], atol=5e-6)


def test_vote_center_persistent_equals_per_workgroup_and_global_paths():
    """Ragged batch large enough for the persistent work-list kernel (B * slabs(cells_cap) >= 256): same grid, argmax
    and centre as the one-item-per-workgroup launch (mode bit 0x800) and the global-atomic path (mode 2)."""
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    Ns, Ts = [900, 2048, 300, 1500, 4096, 700, 1200], [4000, 9000, 1500, 6000, 12000, 2500, 5000]
    B = len(Ns)
    scs = [synth.make_scene(9, b, n) for b, n in enumerate(Ns)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = torch.cat([ops.sample_tuples(n, t, 5, 9, (b,)) for b, (n, t) in enumerate(zip(Ns, Ts))])
    lg = torch.cat([torch.from_numpy(synth.teacher_logits(s["pc_canon"], idx[sum(Ts[:b]):sum(Ts[:b + 1])].cpu().numpy(), 32))
                    for b, s in enumerate(scs)]).to(dev)
    u = torch.cat([ops.philox_uniform(t, 6, 9, 1, (b,)) for b, t in enumerate(Ts)])
    outs = []
    for mode in (0, 0x800, 2):
        pipe = VotingPipeline(Ns, Ts, num_rots=90, vote_mode=mode)
        pipe.decode(pts, idx, lg, u)
        grid = torch.zeros(B * pipe.cells_cap, dtype=torch.int32, device=dev)
        goff = torch.arange(B, dtype=torch.int64, device=dev) * pipe.cells_cap
        pipe.vote_center(pts, idx, grid=grid, grid_off=goff)
        ncell = pipe.grids.cpu().numpy().view(np.int32).reshape(B, 8)[:, 6]
        g = grid.cpu().numpy().reshape(B, -1)
        outs.append((pipe.argmax.cpu().numpy().copy(), pipe.peak.cpu().numpy().copy(), pipe.world.cpu().numpy().copy(),
                     [g[b, :ncell[b]].copy() for b in range(B)]))
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0]) and np.array_equal(outs[0][1], other[1])
        assert np.array_equal(outs[0][2], other[2])
        for a, b_ in zip(outs[0][3], other[3]):
            assert np.array_equal(a, b_)
    assert outs[0][1].min() > 10


def test_shot_describe_refuses_a_workspace_it_was_not_prepared_for():
    """The two-call SHOT form keeps its state (cell tables, frames, neighbour lists) in a per-(device, stream) workspace:
    describe_device raises instead of reading another call's state; separate streams get separate workspaces."""
    from cppf2_amd._lib import CppfError
    sc = synth.make_scene(3, 0, 900)
    pts = torch.as_tensor(sc["pc"]).cuda()
    off = ops._offsets([900], pts.device)
    other = torch.as_tensor(synth.make_scene(3, 1, 500)["pc"]).cuda()
    off2 = ops._offsets([500], pts.device)
    nrm = shot.prepare_device(pts, off, 0.02, 0.02)
    want = shot.describe_device(pts, off, nrm, 0.02)
    shot.compute_device(other, off2, 0.02, 0.02)                  # another SHOT call on the same stream: state is gone
    with pytest.raises(CppfError):
        shot.describe_device(pts, off, nrm, 0.02)
    shot.prepare_device(pts, off, 0.02, 0.02)
    with pytest.raises(CppfError):
        shot.describe_device(pts, off, nrm, 0.015)                # other radius than the one prepared for
    # a second stream has its own workspace: interleaving the halves of two clouds on two streams is fine
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    n1 = shot.prepare_device(pts, off, 0.02, 0.02)
    with torch.cuda.stream(s2):
        n2 = shot.prepare_device(other, off2, 0.02, 0.02)
    a = shot.describe_device(pts, off, n1, 0.02)
    with torch.cuda.stream(s2):
        b = shot.describe_device(other, off2, n2, 0.02)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(want))
    ref_b, _ = shot.compute_device(other, off2, 0.02, 0.02)
    assert torch.allclose(torch.nan_to_num(b), torch.nan_to_num(ref_b), atol=1e-6)


def test_vote_center_64_scene_full_size_batch():
    """BASELINE-size batch (64 scenes x 4096 points x 20 000 tuples x 180 rotations): the throughput configuration of the
    persistent vote kernel (B > 48: one part per slab, no merge area).  Full grids of all 64 scenes equal the global-atomic
    path's (the independent A/B implementation) cell for cell; two scenes are checked against the oracle's int64 grid."""
    from oracle import cppf_oracle as O
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    B, N, T, R = 64, 4096, 20000, 180
    scs = [synth.make_scene(0, b, N) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = ops.sample_tuples(N, T, 5, 0, tuple(range(B)))
    canon = torch.from_numpy(np.concatenate([s["pc_canon"] for s in scs])).to(dev)
    base = (torch.arange(B, device=dev, dtype=torch.int64) * N).repeat_interleave(T)
    coords = canon[(idx[:, :2].long() + base[:, None]).reshape(-1)].reshape(B * T, 6)
    pos = (coords.clamp(-0.5, 0.5) + 0.5) * 31.0
    kb = torch.arange(32, device=dev, dtype=torch.float32)
    lg = (-0.5 * ((kb[None, None, :] - pos[..., None]) / 0.6) ** 2).contiguous()
    u = ops.philox_uniform(T, 6, 0, 1, tuple(range(B)))
    outs = []
    for mode in (0, 2):
        pipe = VotingPipeline([N] * B, [T] * B, num_rots=R, vote_mode=mode, cells_cap=1 << 19)
        pipe.decode(pts, idx, lg, u)
        grid = torch.zeros(B * pipe.cells_cap, dtype=torch.int32, device=dev)
        goff = torch.arange(B, dtype=torch.int64, device=dev) * pipe.cells_cap
        pipe.vote_center(pts, idx, grid=grid, grid_off=goff)
        torch.cuda.synchronize()
        outs.append((pipe.argmax.cpu().numpy().copy(), pipe.peak.cpu().numpy().copy(), pipe.world.cpu().numpy().copy(), grid,
                     pipe.grids.cpu().numpy().view(np.int32).reshape(B, 8).copy(), pipe.tr.cpu().numpy().copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][3], outs[1][3])                       # 64 full grids, cell for cell
    ncell = outs[0][4][:, 6]
    assert ncell.min() > 100000 and ncell.max() <= (1 << 19) and outs[0][1].min() > 500
    g = outs[0][3].cpu().numpy().reshape(B, -1)
    trig = (pipe.cs.cpu().numpy(), pipe.sn.cpu().numpy())
    idx_np = idx.cpu().numpy()
    for b in (0, 37):
        sl = slice(b * T, (b + 1) * T)
        grid_o, cand = O.vote_center(scs[b]["pc"], outs[0][5][sl], 2e-3, idx_np[sl][:, :2], R, trig=trig)
        assert grid_o.size == ncell[b] and np.array_equal(grid_o.reshape(-1), g[b, :ncell[b]].astype(np.int64))
        assert int(np.argmax(grid_o)) == int(outs[0][0][b]) and np.array_equal(cand, outs[0][2][b])


@pytest.mark.parametrize("B,bmm,R", [(20, 100000, 90),     # >= 16 scenes: ~160-pair row blocks, several chunks per scene
                                     (20, 7001, 36),       # chunk boundaries inside pairs (7001 is not a multiple of 36)
                                     (5, 4000, 72),        # small-batch plan (32-pair blocks), many chunks
                                     (3, 100, 90)])        # chunks barely longer than one pair's rotations
def test_rot_bins_chunk_aligned_blocks_equal_the_exhaustive_sweep(B, bmm, R):
    """Both rotation votes from the lookup-table kernel (row blocks aligned to the float32 accumulation chunks, one
    launch for both axes) against the exhaustive sweep and the single-axis entry point on ragged batches: counts are
    float32 sums folded chunk by chunk (eval.py:41-45), so every chunk boundary must fall where the reference puts it."""
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    rng = np.random.RandomState(B * 1000 + R)
    Ns = [int(x) for x in rng.randint(300, 1500, B)]
    Ts = [int(x) for x in rng.randint(800, 9000, B)]
    Ts[1] = 40                                           # a scene with almost nothing kept
    scs = [synth.make_scene(4, b, n) for b, n in enumerate(Ns)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = torch.cat([ops.sample_tuples(n, t, 5, 4, (b,)) for b, (n, t) in enumerate(zip(Ns, Ts))])
    idx[sum(Ts[:2]) + 3, 1] = idx[sum(Ts[:2]) + 3, 0]    # a degenerate pair (|ab| = 0): dropped by vote_rotation
    lg = torch.cat([torch.from_numpy(synth.teacher_logits(s["pc_canon"], idx[sum(Ts[:b]):sum(Ts[:b + 1])].cpu().numpy(), 32))
                    for b, s in enumerate(scs)]).to(dev)
    u = torch.cat([ops.philox_uniform(t, 6, 4, 1, (b,)) for b, t in enumerate(Ts)])
    pipe = VotingPipeline(Ns, Ts, num_rots=R, bmm_size=bmm, backproj_ratio=0.3)
    assert pipe.lut is not None
    pipe.decode(pts, idx, lg, u)
    pipe.vote_center(pts, idx)
    pipe.backvote(pts, idx)
    pipe.rot_bins(pts, idx, use_lut=False)
    dense, dtop = pipe.counts.cpu().numpy().copy(), pipe.top_idx.cpu().numpy().copy()
    pipe.counts.zero_()
    pipe.rot_bins(pts, idx, use_lut=True)
    fused, ftop = pipe.counts.cpu().numpy().copy(), pipe.top_idx.cpu().numpy().copy()
    pipe.counts.zero_()
    pipe.rot_bins_single(pts, idx, 0)
    pipe.rot_bins_single(pts, idx, 1)
    single = pipe.counts.cpu().numpy().copy()
    assert np.array_equal(fused, dense) and np.array_equal(ftop, dtop)
    assert np.array_equal(single, dense)
    kept = pipe.kept_count.cpu().numpy()
    assert kept.min() >= 0 and kept.max() * R > bmm       # at least one scene spans more than one chunk
    assert dense[:, kept > 0].max() > 0


@pytest.mark.parametrize("R", [8, 37, 512, 600])
def test_vote_center_rotation_counts_at_the_table_limits(R):
    """num_rots at and around the LDS rotation-table capacity (512: arcs path with the doubled table; 600: the exhaustive
    path) and odd counts: the grid equals the oracle's int64 grid cell for cell."""
    from oracle import cppf_oracle as O
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    N, T = 1500, 3000
    sc = synth.make_scene(12, 0, N)
    idx = ops.sample_tuples(N, T, 5, 12, (0,))
    lg = torch.from_numpy(synth.teacher_logits(sc["pc_canon"], idx.cpu().numpy(), 32)).to(dev)
    u = ops.philox_uniform(T, 6, 12, 1, (0,))
    pts = torch.from_numpy(sc["pc"]).to(dev)
    pipe = VotingPipeline([N], [T], num_rots=R, vote_mode=1)
    pipe.decode(pts, idx, lg, u)
    grid = torch.zeros(pipe.cells_cap, dtype=torch.int32, device=dev)
    pipe.vote_center(pts, idx, grid=grid, grid_off=torch.zeros(1, dtype=torch.int64, device=dev))
    trig = (pipe.cs.cpu().numpy(), pipe.sn.cpu().numpy())
    want, cand = O.vote_center(sc["pc"], pipe.tr.cpu().numpy(), 2e-3, idx.cpu().numpy()[:, :2], R, trig=trig)
    got = grid.cpu().numpy()[:want.size].astype(np.int64)
    assert np.array_equal(got, want.reshape(-1))
    assert int(pipe.argmax.item()) == int(np.argmax(want)) and np.array_equal(pipe.world.cpu().numpy()[0], cand)


@pytest.mark.gpu
def test_decode_bins_with_masked_logits():
    """-inf logits (masked bins) take no probability mass: the draw lands in the unmasked bins only, no NaN (eval.py:225-229)."""
    import torch
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, T = 64, 500
    pts = torch.randn(N, 3, device=dev) * 0.05
    idx = torch.randint(0, N, (T, 5), device=dev, dtype=torch.int32)
    logits = torch.randn(T, 6, 32, device=dev)
    logits[:, :, :10] = float("-inf")
    logits[:, :, 25:] = float("-inf")
    u = torch.rand(T, 6, device=dev)
    pipe = VotingPipeline([N], [T], num_rots=36)
    pipe.decode(pts, idx, logits, u)
    b = pipe.bins.cpu().numpy()
    assert b.min() >= 10 and b.max() <= 24
    assert bool(torch.isfinite(pipe.tr).all()) and bool(torch.isfinite(pipe.scale).all())


def _vote_grids_both_kernels(pcs, Ts, R, res, tr, seed):
    """The batch's vote grids, first maxima and peaks from the LDS-slab kernel (mode 1) and from the one-atomic-per-vote kernel
    (mode 2: exact divisions, every rotation tested) on the same pairs and vote parameters."""
    from cppf2_amd import _lib
    import ctypes as C
    B, Ns = len(pcs), [len(p) for p in pcs]
    pts = torch.as_tensor(np.concatenate(pcs)).to(DEV)
    idx = torch.as_tensor(np.concatenate([synth.host_sample_tuples(seed, b, Ts[b], 5, Ns[b]) for b in range(B)])).to(DEV, torch.int32)
    out = {}
    for mode in (1, 2):
        pipe = VotingPipeline(Ns, Ts, num_rots=R, res=res, vote_mode=mode)
        pipe.tr.copy_(torch.as_tensor(tr))
        _lib.check(_lib.load().cppf_scene_bounds(B, ops._p(pts), ops._p(pipe.pt_off), C.c_float(pipe.res), ops._p(pipe.grids),
                                                 ops._stream()), "bounds")
        hg = ops.grids_to_host(pipe.grids)
        cells = np.array([g.ncell for g in hg], np.int64)
        off = torch.as_tensor(np.concatenate([[0], np.cumsum(cells)])).to(DEV)
        grid = torch.full((int(cells.sum()),), -1, dtype=torch.int32, device=DEV)
        pipe.vote_center(pts, idx, grid=grid, grid_off=off)
        out[mode] = (grid.cpu().numpy(), pipe.argmax.cpu().numpy().copy(), pipe.peak.cpu().numpy().copy(), cells,
                     [tuple(g.g) for g in hg])
    return out


@pytest.mark.parametrize("shape", ["voxel_cylinders", "thin_plates"])
def test_big_batch_of_small_grids_is_cut_into_fine_slabs(shape):
    """More than 48 scenes whose grids are far smaller than the LDS slab (objects of ~8e4 cells): vote_worklist_kernel cuts them
    into slabs finer than the LDS allows so that the batch still fills the chip.  The grids must stay what the one-atomic-per-vote
    kernel counts, cell for cell, and the first maxima with them.  `thin_plates`: gy * gz exceeds the slab, so x-layer 0 -- never
    a valid cell, train_dino.py:199 -- spans several slabs."""
    B, N, T, R = 56, 1024, 2000, 36
    rng = np.random.default_rng(11)
    if shape == "voxel_cylinders":
        pcs = [synth.make_scene_voxel2mm(7, b, N)["pc"] for b in range(B)]
    else:
        pcs = [(rng.random((N, 3)) * np.array([0.008, 0.4, 0.4]) + np.array([0.1 * b, 0.0, 0.8])).astype(np.float32) for b in range(B)]
    ext = 0.05 if shape == "voxel_cylinders" else 0.2
    tr = np.stack([(rng.random(B * T) - 0.5) * ext, rng.random(B * T) * ext], -1).astype(np.float32)
    out = _vote_grids_both_kernels(pcs, [T] * B, R, 2e-3, tr, 7)
    cells = out[1][3]
    assert cells.max() * B < 36864 * 512                       # the batch is in the fine-slab regime (two items per CU wanted)
    if shape == "thin_plates":
        g = out[1][4][0]
        assert g[1] * g[2] > cells.sum() / 512                 # one x-layer is wider than a slab
    assert out[1][0].min() >= 0 and out[1][0].sum() > (0.2 * B * T * R if shape == "voxel_cylinders" else 1e5)
    assert np.array_equal(out[1][0], out[2][0])
    assert np.array_equal(out[1][1], out[2][1]) and np.array_equal(out[1][2], out[2][2])


@pytest.mark.parametrize("cfg_seed", list(range(6)))
def test_vote_center_slab_cut_random_batches(cfg_seed):
    """Batches drawn from a seed on both sides of the pair-split limit (40 .. 96 scenes): ragged point / pair counts, boxes from
    needles to plates and cubes (grids of 1e2 .. 2e6 cells, so the device-chosen slab ranges from its floor to the LDS capacity), cell sizes,
    rotation counts that are not multiples of the vote quantum.  Slab kernel == one-atomic-per-vote kernel, cell for cell."""
    rng = np.random.default_rng(500 + cfg_seed)
    B = int(rng.integers(40, 97))
    res = float(rng.choice([1.5e-3, 2e-3, 3e-3, 5e-3]))
    R = int(rng.choice([8, 13, 36, 59, 120]))
    big = res * float(rng.choice([25, 50, 80]))         # (the longest box edge is 1.5 x this: grids stay below the 2^21-cell cap)
    pcs, Ts = [], []
    for b in range(B):
        n = int(rng.integers(60, 900))
        box = rng.choice([0.004, 0.02, big], size=3) * (0.5 + rng.random(3))
        pcs.append((rng.random((n, 3)) * box + np.array([0.0, 0.0, 0.6])).astype(np.float32))
        Ts.append(int(rng.integers(100, 1500)))
    tot = sum(Ts)
    tr = np.stack([(rng.random(tot) - 0.5) * big, rng.random(tot) * big], -1).astype(np.float32)
    out = _vote_grids_both_kernels(pcs, Ts, R, res, tr, 900 + cfg_seed)
    assert out[1][3].max() <= 1 << 21 and out[1][0].min() >= 0 and out[1][0].sum() > 0
    assert np.array_equal(out[1][0], out[2][0]), cfg_seed
    assert np.array_equal(out[1][1], out[2][1]) and np.array_equal(out[1][2], out[2][2]), cfg_seed
