"""SHOT352 / normals.  PARITY UNPINNED against PCL (absent from the image, no reference fixtures): the C
oracle (oracle/shot_oracle.c) restates PCL 1.9.1's algorithm and is validated here by invariants and
known-answer clouds on CPU; the HIP kernel is compared with the oracle on the GPU."""
import numpy as np
import pytest

from oracle import shot_oracle as S
from cppf2_amd import synth


def _rot(seed):
    rng = np.random.RandomState(seed)
    q, _ = np.linalg.qr(rng.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_oracle_unit_norm_and_frames():
    sc = synth.make_scene(0, 1, 1500)
    shot, nrm, rf = S.compute(sc["pc"], 0.02, 0.02)
    ok = ~np.isnan(shot).any(1)
    assert ok.mean() > 0.99
    assert np.allclose(np.linalg.norm(shot[ok], axis=1), 1.0, atol=1e-5)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)
    x, y, z = rf[ok, 0:3], rf[ok, 3:6], rf[ok, 6:9]
    assert np.abs((x * z).sum(1)).max() < 1e-5 and np.abs((x * y).sum(1)).max() < 1e-5
    assert np.allclose(np.cross(z, x), y, atol=1e-6)           # y = z x x, right-handed
    # normals face the camera at the origin
    assert np.all((nrm * (-sc["pc"])).sum(1) >= 0)


def test_oracle_plane_known_answer():
    # interior point of a flat patch: every neighbour normal is +-z of the LRF -> only cosine slots 0 / 10
    g = np.arange(-15, 16) * 0.002
    xx, yy = np.meshgrid(g, g)
    pc = np.stack([xx.ravel(), yy.ravel(), np.full(xx.size, 0.8)], -1).astype(np.float32)
    pc += (np.random.RandomState(0).rand(*pc.shape).astype(np.float32) - 0.5) * np.float32([2e-4, 2e-4, 0])
    shot, nrm, rf = S.compute(pc, 0.01, 0.01)
    centre = np.argmin(np.abs(pc[:, :2]).sum(1))
    assert abs(abs(nrm[centre, 2]) - 1) < 1e-5 and nrm[centre, 2] < 0     # faces the origin
    d = shot[centre].reshape(32, 11)
    assert np.all(d[:, 1:10] == 0) and d.sum() > 0
    assert abs(abs(rf[centre, 8]) - 1) < 1e-5


def test_oracle_nan_policy():
    pc = np.array([[0, 0, 1], [0.001, 0, 1], [0, 0.001, 1], [0.5, 0.5, 1.5]], np.float32)
    shot, nrm, rf = S.compute(pc, 0.01, 0.01)
    assert np.isnan(shot).all()                         # < 5 neighbours everywhere
    assert np.isnan(nrm[3]).all() and not np.isnan(nrm[0]).any()   # isolated point: < 3 neighbours


def test_oracle_rigid_motion_invariance():
    # the descriptor depends on geometry relative to the viewpoint only through normal orientation:
    # rotate the cloud about the camera (origin) -> same descriptors
    sc = synth.make_scene(2, 0, 800)
    R = _rot(4)
    s1, n1, _ = S.compute(sc["pc"], 0.02, 0.02)
    s2, n2, _ = S.compute((sc["pc"].astype(np.float64) @ R.T).astype(np.float32), 0.02, 0.02)
    ok = ~(np.isnan(s1).any(1) | np.isnan(s2).any(1))
    # float32 coordinates change under rotation: neighbour sets at the radius edge may differ, and where the
    # two large LRF eigenvalues are (nearly) equal -- interior of the flat caps -- the x axis is ill-conditioned
    # by construction of SHOT, so a minority of descriptors legitimately change
    close = np.abs(s1[ok] - s2[ok]).max(1) < 5e-3
    assert close.mean() > 0.9 and np.median(np.abs(s1[ok] - s2[ok]).max(1)) < 1e-4
    assert np.abs(n1 @ R.T - n2).max(1).mean() < 1e-3


def test_oracle_deviation_from_pcl_arithmetic_is_bounded():
    """The oracle's stated deviations from PCL 1.9.1 (double covariance about the query point, Jacobi, index-ordered sums)
    against its PCL-arithmetic mode (single-pass float covariance on raw coordinates, closed-form eigen33, distance-ordered
    sums) on bench clouds (4096 points, r = 2 cm, ~0.8 m from the camera).  This is the measured size of the deviations:
    SHOT parity stays UNPINNED (no PCL here), but it is bounded by these numbers, printed for DESIGN.md."""
    stats = []
    for sid in (0, 1):
        sc = synth.make_scene(0, sid, 4096)
        s0, n0, rf0, d0 = S.compute_ex(sc["pc"], 0.02, 0.02, pcl_arithmetic=False)
        s1, n1, rf1, d1 = S.compute_ex(sc["pc"], 0.02, 0.02, pcl_arithmetic=True)
        assert np.array_equal(np.isnan(s0), np.isnan(s1))
        ok = ~np.isnan(s0).any(1)
        ang = np.degrees(np.arccos(np.clip((n0 * n1).sum(1), -1, 1)))
        dd = np.abs(s0[ok] - s1[ok]).max(1)
        flips = ((rf0[ok, 0:3] * rf1[ok, 0:3]).sum(1) < 0.5) | ((rf0[ok, 6:9] * rf1[ok, 6:9]).sum(1) < 0.5)
        stats.append(dict(normal_deg_median=float(np.median(ang)), normal_deg_max=float(ang.max()),
                          desc_median=float(np.median(dd)), desc_p90=float(np.percentile(dd, 90)),
                          desc_jump_rows=float((dd > 1e-2).mean()), desc_max=float(dd.max()), lrf_flips=float(flips.mean())))
    print("oracle vs PCL-arithmetic mode:", stats)
    for st in stats:
        # float covariance on raw coordinates at z ~ 0.8 m tilts normals by a few hundredths of a degree
        assert st["normal_deg_median"] < 0.1 and st["normal_deg_max"] < 1.0
        # the LRF (double in both modes) does not move
        assert st["lrf_flips"] == 0.0
        # the descriptor follows continuously (1e-4 level) except where a neighbour's normal crosses one of PCL's cosine
        # steps -- a jump of PCL's own interpolation (see shot_accumulate) that hits a minority of rows
        assert st["desc_median"] < 1e-3 and st["desc_jump_rows"] < 0.15 and st["desc_max"] < 0.5


def test_oracle_modes_agree_when_the_covariance_is_well_conditioned():
    """Same cloud moved to the origin (no cancellation in the float covariance): the two modes give the same normals to
    float rounding -- the deviation measured above is the float covariance's, nothing structural."""
    sc = synth.make_scene(0, 0, 1500)
    pc = (sc["pc"] - sc["pc"].mean(0)).astype(np.float32)
    _, n0, rf0, _ = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=False)
    _, n1, rf1, _ = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=True)
    # (the viewpoint flip may differ for normals perpendicular to the view ray; compare up to sign)
    ang = np.degrees(np.arccos(np.clip(np.abs((n0 * n1).sum(1)), -1, 1)))
    assert np.median(ang) < 2e-3 and np.percentile(ang, 99) < 0.05
    assert np.allclose(rf0, rf1, atol=1e-6, equal_nan=True)


def test_oracle_shot1344_invariants():
    """shot.compute_color (src_shot/shot.cpp:102-161, SHOT1344; parity unpinned like SHOT352): unit norm over all 1344
    entries; the shape channel is SHOT352 up to the common normalisation; both channels put the same mass into every
    spatial sector (they share the interpolation weights); a cloud of one colour fills colour slot 0 only."""
    sc = synth.make_scene(0, 0, 1500)
    rng = np.random.RandomState(0)
    col = rng.rand(1500, 3).astype(np.float32)
    s352, n352, _ = S.compute(sc["pc"], 0.02, 0.02)
    s, nrm, _ = S.compute_color(sc["pc"], col, 0.02, 0.02)
    ok = ~np.isnan(s).any(1)
    assert ok.mean() > 0.99 and np.array_equal(nrm, n352, equal_nan=True)
    assert np.allclose(np.linalg.norm(s[ok], axis=1), 1.0, atol=1e-5)
    shape = s[ok, :352]
    assert np.abs(shape / np.linalg.norm(shape, axis=1, keepdims=True) - s352[ok]).max() < 1e-6
    a, b = shape.reshape(-1, 32, 11).sum(2), s[ok, 352:].reshape(-1, 32, 31).sum(2)
    assert np.abs(a - b).max() < 1e-6
    one, _, _ = S.compute_color(sc["pc"], np.full((1500, 3), 0.3, np.float32), 0.02, 0.02)
    cu = one[ok, 352:].reshape(-1, 32, 31)
    assert np.all(cu[:, :, 1:] == 0) and cu[:, :, 0].sum() > 0
    # colour is stored as uint8 = c * 255 truncated (shot.cpp:114-116): changes below one level do not matter
    s2, _, _ = S.compute_color(sc["pc"], (np.floor(col * 255.0) / 255.0 + 1e-4).astype(np.float32), 0.02, 0.02)
    assert np.array_equal(s2, s, equal_nan=True)


@pytest.mark.gpu
def test_hip_shot_vs_oracle():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import ops, shot
    scs = [synth.make_scene(5, s, n) for s, n in enumerate((1200, 777))]
    pts = torch.as_tensor(np.concatenate([s["pc"] for s in scs])).cuda()
    pt_off = ops._offsets([1200, 777], pts.device)
    hs, hn, hrf = shot.compute_device(pts, pt_off, 0.02, 0.02, want_rf=True, arithmetic="f64")
    hs, hn, hrf = hs.cpu().numpy(), hn.cpu().numpy(), hrf.cpu().numpy()
    o = 0
    for sc in scs:
        n = sc["pc"].shape[0]
        os_, on, orf = S.compute(sc["pc"], 0.02, 0.02)
        assert np.array_equal(np.isnan(os_), np.isnan(hs[o:o + n]))
        # normals / frames: float64 Jacobi with +,-,*,/,sqrt only; partial sums are combined in a different
        # order across lanes (1e-16 relative) -> equal to float32 rounding
        assert np.allclose(hn[o:o + n], on, atol=2e-6, equal_nan=True)
        assert np.allclose(hrf[o:o + n], orf, atol=2e-5, equal_nan=True)
        # histogram: float32 accumulation in a different order + device acos/atan2 ulps
        ok = ~np.isnan(os_).any(1)
        assert np.abs(hs[o:o + n][ok] - os_[ok]).max() < 2e-5
        o += n
    # two-call form == one-call form (same kernels, same workspace)
    n2 = shot.prepare_device(pts, pt_off, 0.02, 0.02, arithmetic="f64")
    s2 = shot.describe_device(pts, pt_off, n2, 0.02)
    assert np.array_equal(n2.cpu().numpy(), hn, equal_nan=True)
    assert np.allclose(s2.cpu().numpy(), hs, atol=1e-6, equal_nan=True)
    # drop-in module call convention (src_shot/shot.cpp:45): list of two flat float32 arrays
    r = shot.compute(scs[1]["pc"].astype(np.float64), 0.02, 0.02, arithmetic="f64")
    assert isinstance(r, list) and r[0].shape == (777 * 352,) and r[1].shape == (777 * 3,) and r[0].dtype == np.float32
    assert np.allclose(r[0].reshape(-1, 352), hs[1200:], atol=1e-6, equal_nan=True)
    nn = shot.estimate_normal(scs[1]["pc"], 0.02, arithmetic="f64")
    assert np.allclose(nn.reshape(-1, 3), hn[1200:], atol=1e-7, equal_nan=True)
    # the module default is PCL's arithmetic for the normals (next test); the two differ by the float covariance's noise
    assert shot.ARITHMETIC == "pcl"
    rp = shot.compute(scs[1]["pc"], 0.02, 0.02)
    ang = np.degrees(np.arccos(np.clip((rp[1].reshape(-1, 3) * hn[1200:]).sum(1), -1, 1)))
    assert 0 < np.nanmedian(ang) < 0.1 and np.nanmax(ang) < 1.0


@pytest.mark.gpu
def test_hip_shot_nan_policy_and_plane():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import shot
    pc = np.array([[0, 0, 1], [0.001, 0, 1], [0, 0.001, 1], [0.5, 0.5, 1.5]], np.float32)
    s, n = shot.compute(pc, 0.01, 0.01)
    assert np.isnan(s).all()
    n = n.reshape(-1, 3)
    assert np.isnan(n[3]).all() and not np.isnan(n[0]).any()


@pytest.mark.gpu
def test_hip_shot_bench_size_vs_oracle():
    """BASELINE-size clouds (4096 points, normal and descriptor radius 2 cm = 10 x res, eval.py:210), a batch of two:
    normals and frames to float rounding; descriptors within 2e-5 except where the oracle itself says a neighbour sits on
    a decision boundary of PCL's interpolation (margin < 1e-6) -- there the float / double evaluation may pick the other
    bin.  (Parity with PCL itself stays unpinned; see test_oracle_deviation_from_pcl_arithmetic_is_bounded.)"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import ops, shot
    scs = [synth.make_scene(0, s, 4096) for s in (0, 1)]
    pts = torch.as_tensor(np.concatenate([s["pc"] for s in scs])).cuda()
    pt_off = ops._offsets([4096, 4096], pts.device)
    hs, hn, hrf = shot.compute_device(pts, pt_off, 0.02, 0.02, want_rf=True, arithmetic="f64")
    hs, hn, hrf = hs.cpu().numpy(), hn.cpu().numpy(), hrf.cpu().numpy()
    for b, sc in enumerate(scs):
        sl = slice(4096 * b, 4096 * (b + 1))
        os_, on, orf, d = S.compute_ex(sc["pc"], 0.02, 0.02)
        assert np.array_equal(np.isnan(os_), np.isnan(hs[sl]))
        assert np.allclose(hn[sl], on, atol=2e-6, equal_nan=True)
        assert np.allclose(hrf[sl], orf, atol=2e-5, equal_nan=True)
        ok = ~np.isnan(os_).any(1)
        assert ok.mean() > 0.99
        err = np.abs(hs[sl][ok] - os_[ok]).max(1)
        exempt = (d[ok, 5] < 1e-6) | (d[ok, 8] < 4e-7)
        assert np.all(err[~exempt] < 2e-5), float(err[~exempt].max())
        assert (err >= 2e-5).mean() < 5e-3
        assert np.allclose(np.linalg.norm(hs[sl][ok], axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_hip_shot_pcl_arithmetic_vs_oracle():
    """The default arithmetic of shot.compute since round 4: pcl::NormalEstimation's (src_shot/shot.cpp:25-32, 66-72 -- single-pass
    float32 sums of the raw coordinates in (distance, index) order, closed-form eigen33, float32 viewpoint flip) against the
    oracle's PCL-arithmetic mode (compute_ex(pcl_arithmetic=True)) at bench size, same criteria as the float64 mode: the
    float32 sums are bit-identical by construction (same addends, same order), what is left in the normals is the
    device's atan2f / cosf / sinf against glibc's inside the closed-form roots; descriptors within 2e-5 except where the oracle
    says a neighbour sits on a decision boundary of PCL's interpolation."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import ops, shot
    scs = [synth.make_scene(0, s, 4096) for s in (0, 1)] + [synth.make_scene(3, 0, 900)]
    sizes = [s["pc"].shape[0] for s in scs]
    pts = torch.as_tensor(np.concatenate([s["pc"] for s in scs])).cuda()
    pt_off = ops._offsets(sizes, pts.device)
    hs, hn, hrf = shot.compute_device(pts, pt_off, 0.02, 0.02, want_rf=True, arithmetic="pcl")
    hs, hn, hrf = hs.cpu().numpy(), hn.cpu().numpy(), hrf.cpu().numpy()
    o = 0
    for sc, n in zip(scs, sizes):
        sl = slice(o, o + n)
        o += n
        os_, on, orf, d = S.compute_ex(sc["pc"], 0.02, 0.02, pcl_arithmetic=True)
        assert np.array_equal(np.isnan(os_), np.isnan(hs[sl])) and np.array_equal(np.isnan(on), np.isnan(hn[sl]))
        ang = np.degrees(np.arccos(np.clip((hn[sl] * on).sum(1), -1, 1)))
        assert np.nanmax(np.abs(hn[sl] - on)) < 2e-5, float(np.nanmax(np.abs(hn[sl] - on)))
        assert np.nanmedian(ang) < 1e-4
        assert np.allclose(hrf[sl], orf, atol=2e-5, equal_nan=True)
        ok = ~np.isnan(os_).any(1)
        err = np.abs(hs[sl][ok] - os_[ok]).max(1)
        # a neighbour normal that moved by 1e-5 may cross a cosine step the oracle's did not: same exemption, by margin
        exempt = (d[ok, 5] < 2e-5) | (d[ok, 8] < 4e-7)
        assert np.all(err[~exempt] < 5e-5), float(err[~exempt].max())
        assert (err >= 5e-5).mean() < 2e-2
    # the normals alone through estimate_normal, and a cloud centred at the origin (well-conditioned sums: both arithmetics agree)
    nn = shot.estimate_normal(scs[2]["pc"], 0.02)
    assert np.allclose(nn.reshape(-1, 3), hn[o - sizes[2]:o], atol=1e-7, equal_nan=True)
    pc0 = (scs[2]["pc"] - scs[2]["pc"].mean(0)).astype(np.float32)
    a = shot.estimate_normal(pc0, 0.02, arithmetic="pcl").reshape(-1, 3)
    b = shot.estimate_normal(pc0, 0.02, arithmetic="f64").reshape(-1, 3)
    ang = np.degrees(np.arccos(np.clip(np.abs((a * b).sum(1)), -1, 1)))
    assert np.nanmedian(ang) < 2e-3 and np.nanpercentile(ang, 99) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["pcl", "f64"])
def test_hip_shot_at_voxel_grid_density_vs_oracle(arith):
    """The density real inputs have (eval.py:185-201 keeps one point per 2 mm voxel: ~250 neighbours inside the 10 x res support,
    bench.py --cloud voxel2mm / cppf2_amd.synth.make_scene_voxel2mm) -- the long-list paths of the kernels: shot_pcl_long's second
    pass for every query (PCL arithmetic), shot_cov's multi-pass float64 sums, shot_hist's four rounds and selection tie-break --
    against the oracle in both arithmetics, same criteria as at bench density."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from scipy.spatial import cKDTree
    from cppf2_amd import ops, shot
    scs = [synth.make_scene_voxel2mm(0, s, 4096) for s in (0, 5)]
    nn = np.array([len(x) for x in cKDTree(scs[0]["pc"]).query_ball_point(scs[0]["pc"], 0.02)])
    assert 200 < nn.mean() < 300 and nn.max() > 256                      # four to five 64-entry rounds per query
    pts = torch.as_tensor(np.concatenate([s["pc"] for s in scs])).cuda()
    pt_off = ops._offsets([4096, 4096], pts.device)
    hs, hn, hrf = shot.compute_device(pts, pt_off, 0.02, 0.02, want_rf=True, arithmetic=arith)
    hs, hn, hrf = hs.cpu().numpy(), hn.cpu().numpy(), hrf.cpu().numpy()
    ntol, dtol, margin = (2e-5, 5e-5, 2e-5) if arith == "pcl" else (2e-6, 2e-5, 1e-6)
    for b, sc in enumerate(scs):
        sl = slice(4096 * b, 4096 * (b + 1))
        os_, on, orf, d = S.compute_ex(sc["pc"], 0.02, 0.02, pcl_arithmetic=(arith == "pcl"))
        assert np.array_equal(np.isnan(os_), np.isnan(hs[sl])) and np.array_equal(np.isnan(on), np.isnan(hn[sl]))
        assert np.nanmax(np.abs(hn[sl] - on)) < ntol, float(np.nanmax(np.abs(hn[sl] - on)))
        assert np.allclose(hrf[sl], orf, atol=2e-5, equal_nan=True)
        ok = ~np.isnan(os_).any(1)
        assert ok.mean() > 0.99
        err = np.abs(hs[sl][ok] - os_[ok]).max(1)
        exempt = (d[ok, 5] < margin) | (d[ok, 8] < 4e-7)
        assert np.all(err[~exempt] < dtol), float(err[~exempt].max())
        assert (err >= dtol).mean() < 2e-2
        assert np.allclose(np.linalg.norm(hs[sl][ok], axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_hip_shot1344_vs_oracle():
    """shot.compute_color on the GPU (cppf_shot1344) against the oracle: rows agree to 2e-5 except where the oracle
    itself has a neighbour within 1e-6 of a decision boundary of PCL's interpolation (shape or colour step)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import shot
    rng = np.random.RandomState(3)
    for sid, n in ((0, 2000), (1, 777)):
        sc = synth.make_scene(6, sid, n)
        col = rng.rand(n, 3).astype(np.float32)
        col[: n // 4] = col[0]                                    # a patch of one colour
        got = shot.compute_color(sc["pc"], col, 0.02, 0.02, arithmetic="f64")       # (the colour oracle runs the float64 normals)
        assert got.shape == (n * 1344,) and got.dtype == np.float32
        got = got.reshape(n, 1344)
        want, _, d = S.compute_color(sc["pc"], col, 0.02, 0.02)
        assert np.array_equal(np.isnan(want), np.isnan(got))
        ok = ~np.isnan(want).any(1)
        err = np.abs(got[ok] - want[ok]).max(1)
        exempt = (d[ok, 5] < 1e-6) | (d[ok, 8] < 4e-7)
        assert np.all(err[~exempt] < 2e-5), float(err[~exempt].max())
        assert (err >= 2e-5).mean() < 1e-2
        assert np.allclose(np.linalg.norm(got[ok], axis=1), 1.0, atol=1e-5)
    # the shape channel is the SHOT352 kernel's
    s352 = shot.compute(sc["pc"], 0.02, 0.02, arithmetic="f64")[0].reshape(n, 352)
    shape = got[ok, :352]
    assert np.abs(shape / np.linalg.norm(shape, axis=1, keepdims=True) - s352[ok]).max() < 1e-5
