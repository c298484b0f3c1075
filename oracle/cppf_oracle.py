"""CPU oracle for the CPPF++ voting hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A NumPy restatement, op for op and dtype for dtype, of the reference's
(qq456cvb/CPPF2) hot-path arithmetic.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package
(cppf2_amd) never does and fails loudly without its HIP library.

Pinning: every function here is checked bit-for-bit (integer outputs, float32
outputs of exact IEEE ops) or to a stated tolerance against outputs of the real
reference functions run in the build container (tests/golden/make_golden.py
imports them from /root/reference; the resulting vectors are committed under
tests/golden/*.npz and replayed by tests/test_oracle_golden.py).

Citations are file:line in /root/reference.

All float32 arithmetic below is written as individually rounded IEEE single
operations, left-to-right sums, and explicit fused multiply-adds in exactly the
three places the torch-CPU reference fuses (torch.norm over 3, torch.cross,
mm with K=3 -- established by probing the reference and recorded next to each
helper).  That reproduces the reference bit-for-bit; the HIP kernels are compiled
with -ffp-contract=off and spell the same fmaf() calls out by hand.
"""
from __future__ import annotations

import math
from itertools import combinations

import numpy as np

F32 = np.float32
F64 = np.float64

# ----------------------------------------------------------------------------
# a1. tuple sampler (eval.py:207 -- np.random.randint(0, N, (T, 2+num_more)))
# The reference draws from NumPy's unseeded global RNG, so "identical inputs"
# can only mean identical *indices*.  The build replaces the draw by a
# counter-based Philox4x32-10 stream keyed by (seed, scene) so the HIP sampler
# and this oracle produce the same table on any machine.
# ----------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-10 (Salmon et al. 2011).  Inputs broadcastable uint32 arrays.
    Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(rounds):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def sample_tuples(seed, scene_id, num_tuples, k, n_points):
    """i.i.d. uniform indices in [0, n_points), shape [T, k] int32.

    Stream definition (shared with csrc/cppf_kernels.hip: sample_tuples_kernel):
      tuple t, column j (j < 8): word j&3 of philox(counter=(t, j>>2, scene, 0),
      key=(seed_lo, seed_hi)); index = (word * N) >> 32   (Lemire range map,
      bias <= N/2^32).
    """
    assert k <= 8
    t = np.arange(num_tuples, dtype=np.uint64)
    seed = int(seed)
    out = np.empty((num_tuples, k), dtype=np.int32)
    for blk in range((k + 3) // 4):
        w = philox4x32(t, np.uint64(blk), np.uint64(scene_id), np.uint64(0),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for j in range(4):
            col = blk * 4 + j
            if col < k:
                out[:, col] = ((w[j].astype(np.uint64) * np.uint64(n_points)) >> np.uint64(32)).astype(np.int32)
    return out


def philox_uniform(seed, scene_id, stream, n, m):
    """[n, m] float32 uniforms in [0,1) (m <= 8): (word >> 8) * 2^-24.
    counter = (row, m_block, scene, stream)."""
    assert m <= 8
    r = np.arange(n, dtype=np.uint64)
    seed = int(seed)
    out = np.empty((n, m), dtype=np.float32)
    for blk in range((m + 3) // 4):
        w = philox4x32(r, np.uint64(blk), np.uint64(scene_id), np.uint64(stream),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for j in range(4):
            col = blk * 4 + j
            if col < m:
                out[:, col] = (w[j] >> np.uint32(8)).astype(np.float32) * F32(2.0 ** -24)
    return out


# ----------------------------------------------------------------------------
# a10. fibonacci_sphere (utils/util.py:191-207); cast to f32 at eval.py:80
# ----------------------------------------------------------------------------
def fibonacci_sphere(samples):
    golden = math.pi * (3.0 - math.sqrt(5.0))          # golden angle
    ys = [1 - (i / float(samples - 1)) * 2 for i in range(samples)]   # +1 ... -1, poles included
    rad = [math.sqrt(1 - y * y) for y in ys]
    return [(math.cos(golden * i) * rad[i], ys[i], math.sin(golden * i) * rad[i]) for i in range(samples)]


def sphere_bins(angle_tol=1.0):
    """eval.py:79-80: num_samples = int(4*pi / (angle_tol/180*pi)); f32 table."""
    num_samples = int(4 * np.pi / (angle_tol / 180 * np.pi))
    return np.array(fibonacci_sphere(num_samples), dtype=np.float32)


# ----------------------------------------------------------------------------
# a3 / a3'. prepare_tuple_inputs
# ----------------------------------------------------------------------------
def tuple_coord_inputs(points, idx):
    """train_shot.py:81 / train_dino.py:92: cat_{i<j}(p_i - p_j) -> [T, C(K,2)*3]."""
    points = np.asarray(points, dtype=F32)
    k = idx.shape[1]
    return np.concatenate([points[idx[:, i]] - points[idx[:, j]]
                           for (i, j) in combinations(range(k), 2)], -1)


def tuple_normal_inputs(normal, idx):
    """train_shot.py:77-79: max(sum(n_i*n_j), sum(-n_i*n_j)) per (i<j) -> [T, C(K,2)].
    torch.sum over the last dim of 3 is a left-to-right f32 sum."""
    normal = np.asarray(normal, dtype=F32)
    k = idx.shape[1]
    cols = []
    for (i, j) in combinations(range(k), 2):
        ni, nj = normal[idx[:, i]], normal[idx[:, j]]
        p = ni * nj
        s = (p[:, 0] + p[:, 1]) + p[:, 2]
        q = (-ni) * nj
        sn = (q[:, 0] + q[:, 1]) + q[:, 2]
        cols.append(np.maximum(s, sn)[:, None])
    return np.concatenate(cols, -1)


def prepare_tuple_inputs_shot(points, idx, feat, normal):
    """train_shot.py:75-83: [coord(30) | normal(10) | feat(K*64)] -> [T, 360]."""
    feat = np.asarray(feat, dtype=F32)
    k = idx.shape[1]
    shot_inputs = np.concatenate([feat[idx[:, i]] for i in range(k)], -1)
    return np.concatenate([tuple_coord_inputs(points, idx),
                           tuple_normal_inputs(normal, idx), shot_inputs], -1)


# ----------------------------------------------------------------------------
# a4. bin decode (eval.py:225-235)
# The reference draws with torch.multinomial on an unseeded generator; the
# oracle (and the HIP kernel) draw by inverse CDF from *supplied* uniforms, which
# is the same distribution.  `margin` lets a test ignore draws whose uniform
# falls within rounding distance of a CDF edge (exp() differs by ulps CPU/GPU).
# ----------------------------------------------------------------------------
def softmax_cdf(pred_cls):
    """Un-normalised softmax terms exp(x - max) (f32), their left-to-right f32 running sum and its total.
    e / total is torch.softmax(pred_cls, -1) of eval.py:228 (pinned by tests/golden/axes.npz: softmax_prob);
    the inverse-CDF draw compares u * total with the running sum, i.e. u with the normalised CDF."""
    pred_cls = np.asarray(pred_cls, dtype=F32)
    NB = pred_cls.shape[-1]
    m = pred_cls.max(-1, keepdims=True)
    e = np.exp((pred_cls - m).astype(F32)).astype(F32)
    cdf = np.empty_like(e)
    acc = np.zeros(pred_cls.shape[:-1], dtype=F32)
    for k in range(NB):                       # left-to-right f32 running sum
        acc = (acc + e[..., k]).astype(F32)
        cdf[..., k] = acc
    return e, cdf, acc


def decode_bins(pred_cls, uniforms, input_pairs, return_margin=False):
    """pred_cls f32[T,6,32] logits, uniforms f32[T,6] in [0,1), input_pairs f32[T,2,3].
    Returns bins int32[T,6], pred_pairs f32[T,2,3], scale f32[T], pred_pairs_scaled f32[T,2,3]."""
    pred_cls = np.asarray(pred_cls, dtype=F32)
    T, C, NB = pred_cls.shape
    e, cdf, acc = softmax_cdf(pred_cls)
    target = (np.asarray(uniforms, dtype=F32) * acc).astype(F32)      # u * total
    bins = (cdf <= target[..., None]).sum(-1).astype(np.int32)        # first k with cdf[k] > target
    bins = np.minimum(bins, NB - 1)
    pred = bins.astype(F32).reshape(T, 2, 3)
    pred = (pred / F32(NB - 1) - F32(0.5)).astype(F32)                # eval.py:230
    ip = np.asarray(input_pairs, dtype=F32)
    d = ip[:, 1] - ip[:, 0]
    real_len = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F32)
    pd = pred[:, 1] - pred[:, 0]
    pred_len = _norm3(pd)                                             # torch.norm (fused), eval.py:234
    scale = (real_len / np.maximum(pred_len, F32(1e-7))).astype(F32)  # eval.py:233-234
    scaled = (pred * scale[:, None, None]).astype(F32)
    if return_margin:
        margin = np.abs(cdf - target[..., None]).min(-1) / acc
        return bins, pred, scale, scaled, margin
    return bins, pred, scale, scaled


# ----------------------------------------------------------------------------
# a5. generate_target_pairs (dataset.py:118-135)
# dtype flow with f32 pairs, int64 axes, f64 centre: pdist/pdist_unit stay f32
# (f32 + python float), everything touching `center` or the axes is f64.
# ----------------------------------------------------------------------------
def generate_target_pairs(point_pairs, up, right, front, center=None):
    """(proj_len, dist2o) of `center` w.r.t. each pair and arccos(u . axis) for three axes.
    Reductions are spelled out the way the HIP kernel computes them (f32 norm of 3 =
    sqrt of a left-to-right sum; f64 3-sums left to right); pinned bit-for-bit against
    the reference function by tests/golden (zero and non-zero centre)."""
    pp = np.asarray(point_pairs, dtype=F32)
    center = np.zeros(3) if center is None else np.asarray(center, dtype=F64)
    a, b = pp[:, 0], pp[:, 1]
    d = a - b
    n = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F32)
    u = (d / (n + F32(1e-7))[:, None]).astype(F32)
    ac = a.astype(F64) - center
    u64 = u.astype(F64)
    p = ac * u64
    proj = (p[:, 0] + p[:, 1]) + p[:, 2]
    oc = ac - proj[:, None] * u64
    dist = np.sqrt((oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1]) + oc[:, 2] * oc[:, 2])
    tr = np.stack([proj, dist], -1).astype(F32)
    rots = []
    for ax in (up, right, front):
        ax = np.asarray(ax, dtype=F64)
        q = u64 * ax
        rots.append(np.arccos((q[:, 0] + q[:, 1]) + q[:, 2]))
    return tr, np.stack(rots, -1).astype(F32)


# ----------------------------------------------------------------------------
# a6. vote_center (train_dino.py:171-215)
# ----------------------------------------------------------------------------
def rotation_table(num_rots):
    """train_dino.py:194: angles = arange(R).float() / R * 2 * pi (f32, left to right);
    cos/sin in f32.  The product (cppf2_amd.ops) builds the same table with torch
    on the host and hands it to the kernels, so table values are an *input* to
    both sides of every parity test."""
    ang = np.arange(num_rots).astype(F32)
    ang = (ang / F32(num_rots)).astype(F32)
    ang = (ang * F32(2)).astype(F32)
    ang = (ang * F32(np.pi)).astype(F32)
    return np.cos(ang).astype(F32), np.sin(ang).astype(F32)


def _fma32(a, b, c):
    """f32 fused multiply-add emulated through f64: the product of two f32 is exact in
    f64; the f64 sum is then rounded once more to f32 (differs from a true FMA only in
    double-rounding ties, ~2^-29 per op)."""
    return (np.asarray(a, F32).astype(F64) * np.asarray(b, F32).astype(F64) + np.asarray(c, F32).astype(F64)).astype(F32)


def _norm3(v):
    """torch.norm(v, dim=-1) for f32 [...,3] on CPU: sqrt(fma(z,z, fma(y,y, x*x))) -- established
    by probing the reference build (0 mismatches on 2e4 vectors; the un-fused form mismatches 11%)."""
    acc = (v[..., 0] * v[..., 0]).astype(F32)
    acc = _fma32(v[..., 1], v[..., 1], acc)
    acc = _fma32(v[..., 2], v[..., 2], acc)
    return np.sqrt(acc).astype(F32)


def _cross(x, u):
    """torch.cross(x, u, dim=-1) for f32 on CPU: each component a*b - c*d is compiled as
    fma(a, b, -(c*d)) (probe: 0 mismatches on 6e5 components; un-fused mismatches 25%)."""
    def fms(a, b, c, d):
        return _fma32(a, b, -(c * d).astype(F32))
    return np.stack([fms(x[:, 1], u[:, 2], x[:, 2], u[:, 1]),
                     fms(x[:, 2], u[:, 0], x[:, 0], u[:, 2]),
                     fms(x[:, 0], u[:, 1], x[:, 1], u[:, 0])], -1).astype(F32)


def _pair_frame(pc, point_idxs):
    a = pc[point_idxs[:, 0]]
    b = pc[point_idxs[:, 1]]
    ab = (a - b).astype(F32)
    nrm = _norm3(ab)
    return a, ab, nrm


def _perp(ab):
    """train_dino.py:187-189: co=(0,-u_z,u_y); if |co|<1e-7: (-u_y,u_x,0)."""
    co = np.stack([np.zeros(ab.shape[0], dtype=F32), -ab[:, 2], ab[:, 1]], -1)
    nco = _norm3(co)
    inv = nco < F32(1e-7)
    if inv.any():
        co[inv] = np.stack([-ab[inv, 1], ab[inv, 0], np.zeros(int(inv.sum()), dtype=F32)], -1)
    return co


def grid_geometry(pc, res):
    """train_dino.py:172-173.  res is a Python float that torch rounds to f32."""
    pc = np.asarray(pc, dtype=F32)
    c0 = pc.min(0)
    c1 = pc.max(0)
    grid_res = ((c1 - c0) / F32(res)).astype(np.int64) + 1
    return c0, grid_res


def vote_center(pc, preds_tr, res, point_idxs, num_rots=36, trig=None, weights=None):
    """Returns (grid_obj int64[gx,gy,gz], cand_world float64[3]) like the reference.
    weights (NOT in the reference; the build's BASELINE-config-5 extension, pinned only by this restatement): per-pair
    weight w in [0,4]; every vote of the pair adds round(w*256) instead of 1."""
    pc = np.asarray(pc, dtype=F32)
    preds_tr = np.asarray(preds_tr, dtype=F32)
    point_idxs = np.asarray(point_idxs)
    res32 = F32(res)
    c0, grid_res = grid_geometry(pc, res)
    proj_len, odist = preds_tr[:, 0], preds_tr[:, 1]
    a, ab, nrm = _pair_frame(pc, point_idxs)
    mask = (nrm > F32(1e-7)) & (odist > res32)
    wfix = None
    if weights is not None:
        w = np.clip(np.asarray(weights, dtype=F32), F32(0), F32(4))
        wfix = ((w * F32(256)).astype(F32) + F32(0.5)).astype(F32).astype(np.int64)[mask]
    proj_len, odist, a, ab, nrm = proj_len[mask], odist[mask], a[mask], ab[mask], nrm[mask]
    ab = (ab / np.maximum(nrm, F32(1e-7))[:, None]).astype(F32)
    c = (a - ab * proj_len[:, None]).astype(F32)
    co = _perp(ab)
    x = ((co / _norm3(co)[:, None]).astype(F32) * odist[:, None]).astype(F32)
    y = _cross(x, ab)
    cs, sn = rotation_table(num_rots) if trig is None else (np.asarray(trig[0], F32), np.asarray(trig[1], F32))
    G = int(grid_res[0] * grid_res[1] * grid_res[2])
    grid = np.zeros(G, dtype=np.int64)
    gr = grid_res
    # chunk over rotations to bound memory
    for r in range(num_rots):
        off = (cs[r] * x).astype(F32) + (sn[r] * y).astype(F32)
        cg = (((c + off).astype(F32) - c0).astype(F32) / res32).astype(F32)
        cg = (cg + F32(0.5)).astype(F32)
        with np.errstate(invalid="ignore"):
            ci = cg.astype(np.int64)                         # trunc toward zero
        valid = np.all(ci > 0, -1) & np.all(ci < gr, -1) & np.all(np.isfinite(cg), -1)
        ci = ci[valid]
        lin = ci[:, 0] * gr[1] * gr[2] + ci[:, 1] * gr[2] + ci[:, 2]
        if wfix is None:
            grid += np.bincount(lin, minlength=G)
        else:
            grid += np.bincount(lin, weights=wfix[valid].astype(np.float64), minlength=G).astype(np.int64)
    grid_obj = grid.reshape(*[int(g) for g in gr])
    cand = np.array(np.unravel_index([np.argmax(grid_obj, axis=None)], grid_obj.shape)).T[::-1][0]
    cand_world = c0 + cand * res                              # f32 + int64*float -> f64
    return grid_obj, cand_world


# ----------------------------------------------------------------------------
# a7. back-vote ("noisy pair") filter + importance weights (eval.py:251-275)
# ----------------------------------------------------------------------------
def backvote_filter(pc, point_idxs_all, targets_tr, up, front, right, T_est,
                    backproj_ratio=0.1, imp_wt_margin=0.01):
    """Returns (pairs_mask bool[T], imp_wt f64[N], imp_pair_wt f64[Tf], back_errs f32[T], thr)."""
    pc = np.asarray(pc, dtype=F32)
    idx = np.asarray(point_idxs_all)
    input_pairs = pc[idx[:, :2]]
    # NOTE the reference passes (up, front, right) into (up, right, front) -- eval.py:252-256
    tr_back, _ = generate_target_pairs(input_pairs, np.array(up), np.array(front), np.array(right), T_est)
    diff = (np.asarray(targets_tr, dtype=F32) - tr_back).astype(F32)
    back_errs = np.linalg.norm(diff, axis=-1)                 # eval.py:257
    thr = np.percentile(back_errs, backproj_ratio * 100)      # eval.py:258
    pairs_mask = back_errs < thr
    flat = idx[pairs_mask, :2].reshape(-1)
    imp_wt = np.bincount(flat, minlength=pc.shape[0]).astype(np.int64)   # eval.py:264-265
    imp_wt = imp_wt / imp_wt.max()                            # eval.py:274 (f64)
    filt = idx[pairs_mask]
    imp_pair_wt = imp_wt[filt[:, :2]].sum(-1) + imp_wt_margin  # eval.py:275
    return pairs_mask, imp_wt, imp_pair_wt, back_errs, thr


# ----------------------------------------------------------------------------
# a8. vote_rotation (train_dino.py:218-239)
# ----------------------------------------------------------------------------
def vote_rotation(pc, preds_rot, point_idxs, num_rots=36, trig=None, tan=None):
    """Returns (up f32[n,R,3], mask bool[T']).  `tan` lets a caller inject tan(preds_rot)."""
    pc = np.asarray(pc, dtype=F32)
    preds_rot = np.asarray(preds_rot, dtype=F32)
    point_idxs = np.asarray(point_idxs)
    a, ab, nrm = _pair_frame(pc, point_idxs)
    mask = nrm > F32(1e-7)
    ab, nrm, preds_rot = ab[mask], nrm[mask], preds_rot[mask]
    ab = (ab / np.maximum(nrm, F32(1e-7))[:, None]).astype(F32)
    co = _perp(ab)
    x = (co / np.maximum(_norm3(co), F32(1e-7))[:, None]).astype(F32)
    y = _cross(x, ab)
    cs, sn = rotation_table(num_rots) if trig is None else (np.asarray(trig[0], F32), np.asarray(trig[1], F32))
    offset = ((cs[None, :, None] * x[:, None]).astype(F32) + (sn[None, :, None] * y[:, None]).astype(F32)).astype(F32)
    tn = np.tan(preds_rot).astype(F32) if tan is None else np.asarray(tan, F32)[mask]
    sgn = np.where(tn > 0, F32(1.0), F32(-1.0)).astype(F32)
    up = ((tn[:, None, None] * offset).astype(F32) + (sgn[:, None, None] * ab[:, None]).astype(F32)).astype(F32)
    n = _norm3(up)
    up = (up / np.maximum(n, F32(1e-7))[..., None]).astype(F32)
    return up, mask


# ----------------------------------------------------------------------------
# a9. get_topk_dir (eval.py:37-51)
# ----------------------------------------------------------------------------
def _dot3_fma(a, b):
    """f32 dot of rows a[M,3] with columns b[3,S] the way an FMA sgemm micro-kernel
    accumulates K=3: fma(a2,b2, fma(a1,b1, a0*b0)).  Products of two f32 are exact
    in f64, so f32(f64 sum) is the fused result except for double-rounding ties."""
    a = a.astype(F64)
    b = b.astype(F64)
    acc = (a[:, 0:1] * b[0:1, :]).astype(F32)
    acc = (acc.astype(F64) + a[:, 1:2] * b[1:2, :]).astype(F32)
    acc = (acc.astype(F64) + a[:, 2:3] * b[2:3, :]).astype(F32)
    return acc


def cone_threshold(angle_tol):
    """eval.py:45: np.cos(2*angle_tol/180*np.pi) compared against an f32 tensor
    -> the comparison happens in f32."""
    return F32(np.cos(2 * angle_tol / 180 * np.pi))


def get_topk_dir(pred, sphere_pts, bmm_size, angle_tol, wt=None, topk=1, return_counts=False, impl="numpy", threads=0):
    """impl="numpy": the definition below.  impl="c": the same arithmetic in oracle/vote_oracle.c (OpenMP, `threads` = 0 for all
    cores; float64 weights only) -- bit-equal counts (tests/test_oracle_golden.py), what bench.py's cpu_baseline times."""
    pred = np.asarray(pred, dtype=F32)
    sphere_pts = np.asarray(sphere_pts, dtype=F32)
    S = sphere_pts.shape[0]
    counts = np.zeros((S,), dtype=F32)
    if wt is None:
        wt = np.ones((pred.shape[0], 1), dtype=F32)
    wt = np.asarray(wt)
    thr = cone_threshold(angle_tol)
    if impl == "c" and wt.dtype == np.float64 and pred.shape[0] > 0:
        from . import vote_oracle as V
        counts = V.sphere_counts(pred, sphere_pts, wt, thr, bmm_size, threads)
        order = np.argsort(-counts, kind="stable")[:topk]
        topk_dir = np.array(sphere_pts[order])
        return (topk_dir, counts[order], counts) if return_counts else (topk_dir, counts[order])
    sph_t = sphere_pts.T
    for i in range((pred.shape[0] - 1) // bmm_size + 1):
        blk = pred[i * bmm_size:(i + 1) * bmm_size]
        w = wt[i * bmm_size:(i + 1) * bmm_size]
        part = np.zeros((S,), dtype=np.result_type(F32, wt.dtype))
        # sub-chunk to bound the [rows, S] temporary; f64 partial sums differ from the
        # reference's single torch.sum only at the 1e-16 level, invisible after the f32 rounding.
        for j in range(0, blk.shape[0], 8192):
            cos = _dot3_fma(blk[j:j + 8192], sph_t)
            hit = (cos > thr).astype(F32)
            part = part + (hit / w[j:j + 8192]).sum(0)
        counts = (counts.astype(part.dtype) + part).astype(F32)
    order = np.argsort(-counts, kind="stable")[:topk]
    topk_dir = np.array(sphere_pts[order])
    if return_counts:
        return topk_dir, counts[order], counts
    return topk_dir, counts[order]


# ----------------------------------------------------------------------------
# a11. pose assembly (eval.py:295-313)
# ----------------------------------------------------------------------------
def assemble_pose(preds_up, preds_right, up, right):
    """Gram-Schmidt of the right vote against the up vote (in the votes' own dtype, f32,
    as the in-place ops at eval.py:295-296 do), third axis by cross product in f64."""
    u = np.array(preds_up)
    r = np.array(preds_right)
    r = r - np.dot(u, r) * u
    r = (r / (np.linalg.norm(r) + 1e-9)).astype(r.dtype)
    iu = int(np.nonzero(np.asarray(up))[0][0])
    ir = int(np.nonzero(np.asarray(right))[0][0])
    io = 3 - iu - ir
    R = np.eye(3)
    R[:, iu] = u
    R[:, ir] = r
    R[:, io] = np.cross(R[:, (io + 1) % 3], R[:, (io + 2) % 3])
    return R


# ----------------------------------------------------------------------------
# whole scene, eval.py:207-313 for one model, with the MLP output injected
# ----------------------------------------------------------------------------
def run_scene(pc, point_idxs_all, pred_cls, pred_scales, uniforms, cfg_up, cfg_right, cfg_front,
              res, num_rots=180, angle_tol=1.0, backproj_ratio=0.1, imp_wt_margin=0.01,
              bmm_size=100000, sphere_pts=None, trig=None, topk_impl="numpy", topk_threads=0):
    """Returns a dict with every stage boundary (used by parity tests and the CPU baseline)."""
    pc = np.asarray(pc, dtype=F32)
    idx = np.asarray(point_idxs_all).astype(np.int64)
    if sphere_pts is None:
        sphere_pts = sphere_bins(angle_tol)
    input_pairs = pc[idx[:, :2]]
    bins, pred_pairs, scale, scaled = decode_bins(pred_cls, uniforms, input_pairs)
    up, front, right = np.array(cfg_up), np.array(cfg_front), np.array(cfg_right)
    targets_tr, targets_rot = generate_target_pairs(scaled, up, front, right)          # eval.py:237-240
    grid_obj, T_est = vote_center(pc, targets_tr, res, idx[:, :2], num_rots, trig)      # eval.py:241
    pairs_mask, imp_wt, imp_pair_wt, back_errs, thr = backvote_filter(
        pc, idx, targets_tr, up, front, right, T_est, backproj_ratio, imp_wt_margin)
    filt = idx[pairs_mask]
    rot_f = targets_rot[pairs_mask]
    out = dict(bins=bins, pred_pairs=pred_pairs, scale=scale, targets_tr=targets_tr, targets_rot=targets_rot,
               grid_obj=grid_obj, argmax=int(np.argmax(grid_obj)), T_est=T_est, pairs_mask=pairs_mask,
               imp_pair_wt=imp_pair_wt, back_errs=back_errs, thr=thr)
    dirs = []
    for col, name in ((0, "up"), (2, "right")):                                         # eval.py:277-293
        cand, vmask = vote_rotation(pc, rot_f[:, col], filt[:, :2], num_rots, trig)
        cand = cand.reshape(-1, 3)
        w = np.broadcast_to(imp_pair_wt[vmask, None], (int(vmask.sum()), num_rots)).reshape(-1, 1)
        d, c, allc = get_topk_dir(cand, sphere_pts, bmm_size, angle_tol, w, topk=1, return_counts=True, impl=topk_impl,
                                  threads=topk_threads)
        out[name + "_idx"] = int(np.argsort(-allc, kind="stable")[0])
        out[name + "_counts"] = allc
        dirs.append(d[0])
    out["preds_up"], out["preds_right"] = dirs[0].copy(), dirs[1].copy()
    out["R_est"] = assemble_pose(dirs[0], dirs[1], up, right)
    if pred_scales is not None:
        ps = np.asarray(pred_scales, dtype=F32)[pairs_mask]
        # torch.median returns the lower median (eval.py:309)
        out["pred_scale"] = np.sort(ps, axis=0)[(ps.shape[0] - 1) // 2]
    return out


# ----------------------------------------------------------------------------
# online alignment refinement, eval.py:319-355  
#
# PARITY UNPINNED: the reference evaluates this block with lietorch 0.2 (SO3.InitFromVec(q).matrix() and its custom
# backward), which is not in /root/reference and not installed here, so no golden vector can be produced.  The
# restatement follows lietorch's published algorithm (lietorch/include/so3.h, groups.py): the rotation of a point
# by the stored quaternion q = (x, y, z, w) is  p + w * 2 (v x p) + v x (2 (v x p))  with v = (x, y, z) and NO
# normalisation of q; matrix() rotates the three basis vectors; the backward of that action hands the group element
# the tangent-space gradient  sum_j (M e_j) x dL/d(M e_j), stored in the first three components of q's gradient
# (the fourth is zero), which eval.py:341 scales by pi/180 before torch.optim.Adam (defaults) adds it to q.
# ----------------------------------------------------------------------------
def so3_matrix(q):
    """lietorch SO3(q).matrix()[:3, :3] for an un-normalised quaternion q = (x, y, z, w), float32."""
    q = np.asarray(q, dtype=F32)
    v, w = q[:3], q[3]
    M = np.empty((3, 3), dtype=F32)
    for j in range(3):
        e = np.zeros(3, dtype=F32)
        e[j] = 1
        uv = np.cross(v, e).astype(F32)
        uv = (uv + uv).astype(F32)
        M[:, j] = (e + w * uv + np.cross(v, uv).astype(F32)).astype(F32)
    return M


def refine_pose(pc, idx2, pred_pairs_scaled, T_est, R_est, y_only, steps=100, lr=1e-2, return_trace=False):
    """eval.py:321-350.  pc f32[N,3]; idx2 int[Tf,2] (kept tuples' first two points); pred_pairs_scaled f32[Tf,2,3];
    returns (T_est f32[3], R_est f32[3,3]) after `steps` Adam updates of the translation and the quaternion."""
    pc = np.asarray(pc, dtype=F32)
    idx2 = np.asarray(idx2).astype(np.int64)
    tgt = np.asarray(pred_pairs_scaled, dtype=F32).reshape(-1, 2, 3)
    t = np.asarray(T_est, dtype=np.float64).astype(F32)            # torch.from_numpy(T_est).cuda().float()
    R0 = np.asarray(R_est, dtype=np.float64).astype(F32)
    q = np.array([0, 0, 0, 1], dtype=F32)
    b1, b2, eps = F32(0.9), F32(0.999), F32(1e-8)
    m_t, v_t = np.zeros(3, F32), np.zeros(3, F32)
    m_q, v_q = np.zeros(4, F32), np.zeros(4, F32)
    P = pc[idx2]                                                   # [Tf,2,3]
    n_el = tgt.shape[0] * 2 * (1 if y_only else 3)
    trace = []
    for step in range(1, steps + 1):
        M = so3_matrix(q)
        rot = (M @ R0).astype(F32)
        d = (P - t).astype(F32)
        c = (d @ rot).astype(F32)                                  # pc_canon[point_idxs]
        r = (c - tgt).astype(F32)
        g = np.sign(r).astype(F32) / F32(n_el)                     # d mean|r| / d c
        if y_only:
            g[..., 0] = 0
            g[..., 2] = 0
            loss = float(np.abs(r[..., 1]).mean())
        else:
            loss = float(np.abs(r).mean())
        trace.append(loss)
        g2, d2 = g.reshape(-1, 3), d.reshape(-1, 3)
        g_t = (-(g2 @ rot.T).sum(0)).astype(F32)                   # c = (p - t) rot
        g_rot = (d2.T @ g2).astype(F32)
        g_M = (g_rot @ R0.T).astype(F32)                           # rot = M R0
        g_xi = np.zeros(3, F32)
        for j in range(3):
            g_xi += np.cross(M[:, j], g_M[:, j]).astype(F32)
        g_q = np.concatenate([g_xi, np.zeros(1, F32)]).astype(F32) / F32(180) * F32(np.pi)
        bc1 = F32(1) - b1 ** F32(step)
        bc2 = F32(1) - b2 ** F32(step)
        for p_, g_, m_, v_ in ((t, g_t, m_t, v_t), (q, g_q, m_q, v_q)):
            m_[:] = b1 * m_ + (F32(1) - b1) * g_
            v_[:] = b2 * v_ + (F32(1) - b2) * g_ * g_
            denom = np.sqrt(v_) / np.sqrt(bc2) + eps
            p_[:] = p_ - (F32(lr) / bc1) * m_ / denom
    R_new = (so3_matrix(q) @ R0).astype(F32)
    if return_trace:
        return t, R_new, trace
    return t, R_new


# ----------------------------------------------------------------------------
# DINO-branch feature plumbing, dataset.py:40-59
# ----------------------------------------------------------------------------
def interpolate_features(descriptors, pts, strides=8, normalize=True):
    """interpolate_features(descriptors[1,C,h,w], pts[n,2] (x, y) pixels) -> [n, C]: F.grid_sample(bilinear, zeros
    padding, align_corners=False) at the pixel centres, then F.normalize over the channels (dataset.py:40-59;
    returned keypoint-major like DINOV2.forward does, dataset.py:79-80).  float32 like torch."""
    d = np.asarray(descriptors, dtype=F32)
    d = d[0] if d.ndim == 4 else d
    C, h, w = d.shape
    p = np.asarray(pts, dtype=F32).reshape(-1, 2)
    gx = ((p[:, 0] + F32(0.5)) / F32(w) / F32(strides)) * F32(2) - F32(1)            # dataset.py:46-47
    gy = ((p[:, 1] + F32(0.5)) / F32(h) / F32(strides)) * F32(2) - F32(1)
    ix = ((gx + F32(1)) * F32(w) - F32(1)) / F32(2)                                   # grid_sampler_unnormalize
    iy = ((gy + F32(1)) * F32(h) - F32(1)) / F32(2)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wts = (((x1 - ix) * (y1 - iy), x0, y0), ((ix - x0) * (y1 - iy), x1, y0),
           ((x1 - ix) * (iy - y0), x0, y1), ((ix - x0) * (iy - y0), x1, y1))          # nw, ne, sw, se
    out = np.zeros((p.shape[0], C), dtype=F32)
    for wgt, xs, ys in wts:
        ok = (xs >= 0) & (xs <= w - 1) & (ys >= 0) & (ys <= h - 1)
        xi = np.clip(xs, 0, w - 1).astype(np.int64)
        yi = np.clip(ys, 0, h - 1).astype(np.int64)
        val = d[:, yi, xi].T                                                           # [n, C]
        out += np.where(ok[:, None], val * wgt.astype(F32)[:, None], F32(0)).astype(F32)
    if normalize:
        nrm = np.sqrt((out.astype(F32) ** 2).sum(1, dtype=F32)).astype(F32)
        out = out / np.maximum(nrm, F32(1e-12))[:, None]
    return out.astype(F32)


# ----------------------------------------------------------------------------
# steps in front of the path, utils/util.py:2586-2607 and 39-46
# ----------------------------------------------------------------------------
def backproject(depth, intrinsics, instance_mask):
    """Masked depth -> points, same convention as utils/util.py:2586-2607: returns points with x and y NEGATED
    (every caller negates them back, eval.py:187-188) and the (row, col) index arrays of the used pixels."""
    depth = np.asarray(depth)
    valid = np.logical_and(instance_mask, depth > 0)
    rows, cols = np.nonzero(valid)
    z = depth[rows, cols]
    pix = np.stack([cols, rows, np.ones_like(cols)], 0).astype(np.float64)
    rays = (np.linalg.inv(intrinsics) @ pix).T
    pts = rays * z[:, None] / rays[:, -1:]
    pts[:, 0] = -pts[:, 0]
    pts[:, 1] = -pts[:, 1]
    return pts, (rows, cols)


def downsample(pc, res, rng=None):
    """One randomly chosen point per `res` voxel (utils/util.py:39-46 does this with open3d's
    voxel_down_sample_and_trace + np.random.choice); voxels are anchored at the cloud's min bound.
    Returns the indices of the kept points, ordered by voxel key."""
    pc = np.asarray(pc)
    rng = np.random if rng is None else rng
    key = np.floor((pc - pc.min(0)) / res).astype(np.int64)
    dims = key.max(0) + 1
    flat = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    order = np.argsort(flat, kind="stable")
    sf = flat[order]
    starts = np.flatnonzero(np.r_[True, sf[1:] != sf[:-1]])
    counts = np.diff(np.r_[starts, len(sf)])
    pick = starts + (rng.random_sample(len(starts)) * counts).astype(np.int64)
    return order[pick]
