"""Seeded synthetic scenes for benchmarks and full-size parity tests.

Host-side input generation only (NumPy); nothing here is on the hot path.
A self-contained counter-based RNG (Philox4x32-10) makes scene `s` of seed `S`
identical on every machine and independent of how scenes are sharded over ranks
(SURVEY.md section 8d/8e): the GPU box regenerates the inputs that the committed
golden summaries (tests/golden) were computed on without needing the reference.

Scene = a bottle-like closed cylinder (radius 0.04 m, height 0.20 m, canonical
"up" = +y as in config/config.yaml) at z ~ 0.8 m in front of the camera with a
random tilt, sampled at N surface points; ground truth pose (R, t) and the NOCS
normalisation (bbox diagonal) come with it so that a "teacher" logit prior can
stand in for trained weights (none ship with the reference: .MISSING_LARGE_BLOBS).
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(ctr, key):
    """ctr: 4 broadcastable arrays of uint32 values, key: (k0, k1) ints -> 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & _LO for x in ctr]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        a = _M0 * c[0]
        b = _M1 * c[2]
        c = [(b >> _S32) ^ c[1] ^ np.uint64(k0), b & _LO, (a >> _S32) ^ c[3] ^ np.uint64(k1), a & _LO]
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def uniforms(seed, scene_id, stream, n, m):
    """float64 [n, m] uniforms in [0,1) with 24-bit resolution; counter = (row, block, scene, stream).
    The float32 view of these values is what cppf_philox_uniform() produces on the device."""
    seed = int(seed)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    rows = np.arange(n, dtype=np.uint64)
    out = np.empty((n, m), dtype=np.float64)
    for blk in range((m + 3) // 4):
        w = philox4x32_10((rows, blk, scene_id, stream), key)
        for j in range(4):
            if blk * 4 + j < m:
                out[:, blk * 4 + j] = (w[j] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    return out


def host_sample_tuples(seed, scene_id, num_tuples, k, n_points):
    """Host mirror of cppf_sample_tuples (same Philox stream, stream id 0): int32 [T, k]."""
    seed = int(seed)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    rows = np.arange(num_tuples, dtype=np.uint64)
    out = np.empty((num_tuples, k), dtype=np.int32)
    for blk in range((k + 3) // 4):
        w = philox4x32_10((rows, blk, scene_id, 0), key)
        for j in range(4):
            if blk * 4 + j < k:
                out[:, blk * 4 + j] = ((w[j].astype(np.uint64) * np.uint64(n_points)) >> _S32).astype(np.int32)
    return out


RADIUS = 0.04
HEIGHT = 0.20
DIAG = float(np.sqrt((2 * RADIUS) ** 2 * 2 + HEIGHT ** 2))   # NOCS normaliser: bbox diagonal


def _rodrigues(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def scene_pose(seed, scene_id, max_tilt_deg=25.0):
    u = uniforms(seed, scene_id, 2, 1, 8)[0]
    tilt_axis = np.array([np.cos(2 * np.pi * u[0]), 0.0, np.sin(2 * np.pi * u[0])])
    R = _rodrigues(tilt_axis, np.deg2rad(max_tilt_deg) * u[1]) @ _rodrigues(np.array([0.0, 1.0, 0.0]), 2 * np.pi * u[2])
    t = np.array([-0.1 + 0.2 * u[3], -0.1 + 0.2 * u[4], 0.7 + 0.2 * u[5]])
    return R, t


def make_scene(seed, scene_id, n_points=4096, max_tilt_deg=25.0):
    """Returns dict(pc f32[N,3], pc_canon f32[N,3] in [-0.5,0.5]^3, R f64[3,3], t f64[3], diag, extent f64[3] =
    the box extents in metres)."""
    u = uniforms(seed, scene_id, 1, n_points, 4)
    side_frac = (2 * np.pi * RADIUS * HEIGHT) / (2 * np.pi * RADIUS * HEIGHT + 2 * np.pi * RADIUS ** 2)
    on_side = u[:, 0] < side_frac
    ang = 2 * np.pi * u[:, 1]
    rad = np.where(on_side, RADIUS, RADIUS * np.sqrt(u[:, 2]))
    top = u[:, 3] < 0.5
    y = np.where(on_side, (u[:, 2] - 0.5) * HEIGHT, np.where(top, HEIGHT / 2, -HEIGHT / 2))
    obj = np.stack([rad * np.cos(ang), y, rad * np.sin(ang)], -1)
    R, t = scene_pose(seed, scene_id, max_tilt_deg)
    pc = (obj @ R.T + t).astype(np.float32)
    # canonical coordinates exactly as eval.py:358 maps them back: (pc - t) @ R / diag
    pc_canon = ((pc.astype(np.float64) - t) @ R / DIAG).astype(np.float32)
    return dict(pc=pc, pc_canon=pc_canon, R=R, t=t, diag=DIAG, extent=np.array([2 * RADIUS, HEIGHT, 2 * RADIUS]))


VOXEL_RADIUS = 0.03
VOXEL_HEIGHT = 0.10


def make_scene_voxel2mm(seed, scene_id, n_points=4096, res=2e-3, max_tilt_deg=25.0):
    """A scene at the point density the reference's pre-processing gives real inputs: eval.py:185-201 keeps one depth pixel per 2 mm
    voxel, so neighbouring points are ~`res` apart and the 10 x res SHOT / normal support (eval.py:210) holds ~250 of them, not the
    ~90 of make_scene()'s uniform surface samples.  The side of a cylinder (radius 0.03 m, height 0.10 m: ~4 700 surface cells of
    res x res for the default 4096 points) under the scene's pose, every point snapped to the `res` lattice of the camera frame and
    jittered by +-0.1 res (a voxel keeps one of its real points, not its centre).  Same keys as make_scene()."""
    u = uniforms(seed, scene_id, 3, n_points, 8)
    ang = 2 * np.pi * u[:, 0]
    obj = np.stack([VOXEL_RADIUS * np.cos(ang), (u[:, 1] - 0.5) * VOXEL_HEIGHT, VOXEL_RADIUS * np.sin(ang)], -1)
    R, t = scene_pose(seed, scene_id, max_tilt_deg)
    world = obj @ R.T + t
    pc = (np.round(world / res) * res + (u[:, 2:5] - 0.5) * (0.2 * res)).astype(np.float32)
    diag = float(np.sqrt((2 * VOXEL_RADIUS) ** 2 * 2 + VOXEL_HEIGHT ** 2))
    pc_canon = ((pc.astype(np.float64) - t) @ R / diag).astype(np.float32)
    return dict(pc=pc, pc_canon=pc_canon, R=R, t=t, diag=diag, extent=np.array([2 * VOXEL_RADIUS, VOXEL_HEIGHT, 2 * VOXEL_RADIUS]))


def teacher_logits(pc_canon, idx, num_bins=32, sigma_bins=0.6, noise=None):
    """Logit prior peaked at the true canonical coordinates of each tuple's first two points
    (layout [T, 6, num_bins] = 2 points x xyz, eval.py:225; bin value = k/(B-1) - 0.5, eval.py:230)."""
    coords = pc_canon[idx[:, :2]].reshape(idx.shape[0], 6).astype(np.float32)
    if noise is not None:
        coords = coords + noise.astype(np.float32)
    pos = (np.clip(coords, -0.5, 0.5) + 0.5) * (num_bins - 1)
    k = np.arange(num_bins, dtype=np.float32)
    return (-0.5 * ((k[None, None, :] - pos[..., None]) / sigma_bins) ** 2).astype(np.float32)
