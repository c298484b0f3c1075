"""Online alignment refinement (eval.py:319-355).  PARITY UNPINNED: lietorch is not available, so the
oracle's restatement of its published algorithm is checked for self-consistency on the CPU (tangent gradient against
autograd, descent from a perturbed pose) and the HIP kernel is checked against that oracle on the GPU."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import cppf_oracle as O   # noqa: E402


def _problem(seed=0, N=600, Tf=400, noise=0.002):
    rng = np.random.RandomState(seed)
    canon = (rng.rand(N, 3).astype(np.float32) - 0.5) * np.array([0.08, 0.2, 0.08], np.float32)
    R_gt = np.linalg.qr(rng.randn(3, 3))[0]
    if np.linalg.det(R_gt) < 0:
        R_gt[:, 0] = -R_gt[:, 0]
    t_gt = np.array([0.05, -0.03, 0.9])
    pc = (canon.astype(np.float64) @ R_gt.T + t_gt).astype(np.float32)
    idx = rng.randint(0, N, (Tf, 2))
    tgt = (canon[idx] + rng.randn(Tf, 2, 3).astype(np.float32) * noise).astype(np.float32)
    return pc, idx, tgt, R_gt, t_gt


def _perturbed(R_gt, t_gt, deg=3.0, shift=0.01):
    a = np.radians(deg)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    return Rz @ R_gt, t_gt + np.array([shift, -shift, shift])


def test_so3_matrix_is_the_rotation_for_unit_quaternions_and_lietorch_formula_otherwise():
    q = np.array([0.1, -0.2, 0.3, 0.9], np.float32)
    qn = q / np.linalg.norm(q)
    M = O.so3_matrix(qn).astype(np.float64)
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(M) - 1) < 1e-6
    x, y, z, w = qn.astype(np.float64)
    want = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert np.allclose(M, want, atol=1e-6)
    v = q[:3].astype(np.float64)
    hat = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    assert np.allclose(O.so3_matrix(q), np.eye(3) + 2 * q[3] * hat + 2 * hat @ hat, atol=1e-6)   # not normalised


def test_oracle_gradients_match_autograd_of_a_left_tangent_perturbation():
    pc, idx, tgt, R_gt, t_gt = _problem(1)
    R0, t0 = _perturbed(R_gt, t_gt)
    q = np.array([0.02, -0.01, 0.03, 1.0], np.float32)
    M = torch.tensor(O.so3_matrix(q), dtype=torch.float64)
    xi = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    t = torch.tensor(t0, requires_grad=True)
    z = xi[0] * 0
    hat = torch.stack([torch.stack([z, -xi[2], xi[1]]), torch.stack([xi[2], z, -xi[0]]), torch.stack([-xi[1], xi[0], z])])
    rot = (torch.eye(3, dtype=torch.float64) + hat) @ M @ torch.tensor(R0)
    c = (torch.tensor(pc[idx], dtype=torch.float64) - t) @ rot
    (c - torch.tensor(tgt, dtype=torch.float64)).abs().mean().backward()
    Mn = O.so3_matrix(q).astype(np.float64)
    rotn = Mn @ R0
    d = (pc[idx] - t0).reshape(-1, 3)
    g = np.sign(d @ rotn - tgt.reshape(-1, 3)) / tgt.size
    g_t = -(g @ rotn.T).sum(0)
    g_M = (d.T @ g) @ R0.T
    g_xi = sum(np.cross(Mn[:, j], g_M[:, j]) for j in range(3))
    assert np.allclose(g_t, t.grad.numpy(), atol=1e-12) and np.allclose(g_xi, xi.grad.numpy(), atol=1e-12)


@pytest.mark.parametrize("y_only", [False, True])
def test_oracle_refinement_descends_from_a_perturbed_pose(y_only):
    pc, idx, tgt, R_gt, t_gt = _problem(2)
    R0, t0 = _perturbed(R_gt, t_gt)
    T, R, trace = O.refine_pose(pc, idx, tgt, t0, R0, y_only, return_trace=True)
    assert len(trace) == 100 and np.mean(trace[-10:]) < 0.6 * trace[0]
    assert np.linalg.norm(T - t_gt) < np.linalg.norm(t0 - t_gt)
    zero = O.refine_pose(pc, idx, tgt, t0, R0, y_only, steps=0)
    assert np.allclose(zero[0], t0.astype(np.float32)) and np.allclose(zero[1], R0.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("y_only", [True, False])
def test_hip_refinement_matches_the_oracle(y_only):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cppf2_amd import ops, synth
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda")
    B, N, T = 3, 2048, 12000
    scs = [synth.make_scene(6, b, N) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([s["pc"] for s in scs])).to(dev)
    idx = ops.sample_tuples(N, T, 5, 6, tuple(range(B)))
    lg = torch.cat([torch.from_numpy(synth.teacher_logits(s["pc_canon"], idx[b * T:(b + 1) * T].cpu().numpy(), 32))
                    for b, s in enumerate(scs)]).to(dev)
    u = ops.philox_uniform(T, 6, 6, 1, tuple(range(B)))
    pipe = VotingPipeline([N] * B, [T] * B, num_rots=72)
    before = pipe.results_to_numpy(pipe.vote(pts, idx, lg, u))
    pipe.refine(pts, idx, y_only)
    after = pipe.results_to_numpy()
    kept_tuple, kept_count = pipe.kept_tuple.cpu().numpy(), pipe.kept_count.cpu().numpy()
    scaled, idx_h = pipe.scaled.cpu().numpy(), idx.cpu().numpy()
    for b in range(B):
        rows = b * T + kept_tuple[b * T: b * T + kept_count[b]]
        want_t, want_R, trace = O.refine_pose(scs[b]["pc"], idx_h[rows][:, :2], scaled[rows], before["t"][b],
                                              before["R"][b].reshape(3, 3), y_only, return_trace=True)
        assert after["flags"][b] & 8 and not before["flags"][b] & 8
        # float32 Adam trajectories: summation order differs (torch / NumPy / wavefront tree), tolerance 0.2 mm, 2e-3
        assert np.abs(after["t"][b] - want_t).max() < 2e-4, (after["t"][b], want_t)
        assert np.abs(after["R"][b].reshape(3, 3) - want_R).max() < 2e-3
        assert np.abs(after["t"][b] - before["t"][b]).max() > 1e-5            # it did move
        # and it is a refinement: the alignment loss it minimises went down
        assert np.mean(trace[-10:]) <= trace[0] + 1e-6
    # steps = 0 leaves the records untouched
    pipe.vote(pts, idx, lg, u)
    pipe.refine(pts, idx, y_only, steps=0)
    same = pipe.results_to_numpy()
    assert np.array_equal(same["t"], before["t"]) and np.array_equal(same["R"], before["R"])
