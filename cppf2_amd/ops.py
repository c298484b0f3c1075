"""Host-side mirror of the reference's operator interface for the voting hot path.

Same names, argument meaning and return types as the reference's Python callables
(train_dino.py:171-239 vote_center / vote_rotation, eval.py:37-51 get_topk_dir,
dataset.py:118-135 generate_target_pairs, utils/util.py:191-207 fibonacci_sphere), on
top of the C ABI in include/cppf_hip.h.  Every function here runs on the GPU through
libcppf_hip.so; there is no CPU path (import fails loudly without the library, calls
fail loudly without a GPU).

PyTorch is plumbing only: device memory, the current HIP stream, dtype conversion.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import CppfError, SceneGrid, SceneResult

_L = _lib.load()

BMM_SIZE = 100000          # eval.py:81


# ----------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------
def _dev():
    if not torch.cuda.is_available():
        raise CppfError("cppf2_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


DYNAMIC_BLOCKS = True        # the cppf_reslayer_split* launches claim their row blocks from a counter (False: fixed shares)
_SCHED = {}                  # (device, stream handle) -> int32[2] scheduling counters of that stream's launches
SCHED_CACHE_MAX = 64


def _sched():
    """The `sched` argument of the cppf_reslayer_split* entry points for a launch on the current stream: two zeroed int32 words
    per (device, stream) -- launches of one stream cannot overlap and every launch leaves them zero, launches of different
    streams must not share them (include/cppf_hip.h).  NULL when DYNAMIC_BLOCKS is off."""
    if not DYNAMIC_BLOCKS:
        return C.c_void_p(0)
    st = torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    buf = _SCHED.get(key)
    if buf is None:
        if len(_SCHED) >= SCHED_CACHE_MAX:             # a process that keeps creating streams does not pin a buffer per stream
            _SCHED.pop(next(iter(_SCHED)))
        buf = _SCHED[key] = torch.zeros((2,), dtype=torch.int32, device=st.device)
    return C.c_void_p(buf.data_ptr())


def _t(x, dtype, device=None):
    """torch tensor on the device with the given dtype, contiguous (accepts numpy / torch / lists)."""
    device = device or _dev()
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), device="cpu").to(device=device, dtype=dtype).contiguous()


def _axes9(a1, a2, a3):
    arr = (C.c_double * 9)(*[float(v) for ax in (a1, a2, a3) for v in np.asarray(ax).reshape(3)])
    return arr


def fibonacci_sphere(samples):
    """utils/util.py:191-207 (host, float64 Python math; cast to float32 by the caller, eval.py:80)."""
    ga = math.pi * (3.0 - math.sqrt(5.0))
    out = []
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        rad = math.sqrt(1 - y * y)
        out.append((math.cos(ga * i) * rad, y, math.sin(ga * i) * rad))
    return out


_SPHERE_CACHE = {}


def sphere_bins(angle_tol=1.0):
    """eval.py:79-80: float32 [int(4*pi/(angle_tol/180*pi)), 3]."""
    n = int(4 * np.pi / (angle_tol / 180 * np.pi))
    if n not in _SPHERE_CACHE:
        _SPHERE_CACHE[n] = np.array(fibonacci_sphere(n), dtype=np.float32)
    return _SPHERE_CACHE[n]


def is_fibonacci(sphere_pts):
    s = np.asarray(sphere_pts)
    if s.ndim != 2 or s.shape[1] != 3 or s.shape[0] < 16:
        return False
    return bool(np.array_equal(s.astype(np.float32), np.array(fibonacci_sphere(s.shape[0]), dtype=np.float32)))


LUT_ROWS, LUT_COLS, LUT_K = 256, 256, 8
_LUT_CACHE = {}


def build_bin_lut(sphere_pts, cos_thr, rows=LUT_ROWS, cols=LUT_COLS):
    """Cell -> candidate-bin table for cppf_rot_bins (host, NumPy, once per bin set).

    The sphere is cut into rows of equal height in y and `cols` azimuth sectors; cell (i, j) lists every bin whose
    cone {v : v.bin > cos_thr} can contain a direction of the cell: bins within (cone angle + circumradius of the
    cell + 2e-3 rad) of the cell centre.  The kernel then tests only those bins exactly, so the counts are the ones
    of the exhaustive sweep for ANY bin set.  Returns int16 [rows, cols, LUT_K] (-1 = empty) or None if some cell
    needs more than LUT_K slots (wide cones): the caller then uses the exhaustive kernel."""
    sph = np.asarray(sphere_pts, dtype=np.float64)
    sph = sph / np.maximum(np.linalg.norm(sph, axis=1, keepdims=True), 1e-30)
    alpha = float(np.arccos(np.clip(cos_thr, -1.0, 1.0)))
    i = np.arange(rows)
    y_hi, y_lo = 1 - i / (rows / 2), 1 - (i + 1) / (rows / 2)
    yc = 0.5 * (y_hi + y_lo)
    dphi = 2 * np.pi / cols
    phic = (np.arange(cols) + 0.5) * dphi

    def vec(y, phi):
        r = np.sqrt(np.maximum(0.0, 1 - y * y))
        return np.stack([r * np.cos(phi), y + 0 * phi, r * np.sin(phi)], -1)
    c0 = vec(yc, phic[0])
    rho = np.zeros(rows)
    for yy in (y_lo, y_hi):
        for dp in (-dphi / 2, dphi / 2):
            rho = np.maximum(rho, np.arccos(np.clip((vec(yy, phic[0] + dp) * c0).sum(-1), -1, 1)))
    rho[0] = max(rho[0], np.arccos(np.clip(yc[0], -1, 1)))          # the polar cells contain the pole
    rho[-1] = max(rho[-1], np.arccos(np.clip(-yc[-1], -1, 1)))
    lim = np.cos(np.minimum(np.pi, alpha + 1.02 * rho + 2e-3))
    masks = []
    for r in range(rows):
        masks.append(vec(np.full(cols, yc[r]), phic) @ sph.T >= lim[r])
    kmax = max(int(m.sum(1).max()) for m in masks)
    if kmax > LUT_K:
        return None
    lut = np.full((rows, cols, LUT_K), -1, dtype=np.int16)
    for r, m in enumerate(masks):
        cc, ss = np.nonzero(m)
        slot = np.zeros(cols, dtype=np.int64)
        for c, sidx in zip(cc, ss):
            lut[r, c, slot[c]] = sidx
            slot[c] += 1
    return lut


def bin_lut_device(sphere_pts, cos_thr, device):
    """Cached device copy of build_bin_lut (None if the table does not fit LUT_K)."""
    sph = np.ascontiguousarray(sphere_pts, dtype=np.float32)
    key = (sph.tobytes(), float(cos_thr), str(device))
    if key not in _LUT_CACHE:
        lut = build_bin_lut(sph, cos_thr)
        _LUT_CACHE[key] = None if lut is None else torch.from_numpy(lut).to(device)
    return _LUT_CACHE[key]


_TRIG_CACHE = {}


def rotation_table(num_rots, device=None):
    """train_dino.py:194-195 built with torch on the HOST (so the table is the one the CPU reference uses),
    uploaded once per (num_rots, device).  Returns (cos, sin) device float32 tensors."""
    device = device or _dev()
    key = (int(num_rots), str(device))
    if key not in _TRIG_CACHE:
        angles = torch.arange(num_rots).float() / num_rots * 2 * np.pi
        _TRIG_CACHE[key] = (torch.cos(angles).to(device), torch.sin(angles).to(device))
    return _TRIG_CACHE[key]


def _trig(num_rots, trig, device):
    if trig is None:
        return rotation_table(num_rots, device)
    return _t(trig[0], torch.float32, device), _t(trig[1], torch.float32, device)


def cone_threshold(angle_tol):
    """eval.py:45 compares an f32 tensor with np.cos(2*angle_tol/180*pi): the comparison happens in f32."""
    return float(np.float32(np.cos(2 * angle_tol / 180 * np.pi)))


def percentile_params(n, ratio):
    """(index k, weight gamma) np.percentile(x_f32[n], ratio*100) interpolates with -- NumPy 2.x arithmetic:
    q = f32(q)/f32(100); virtual = (n-1)*q in f32; gamma = f32(f64(virtual) - floor)."""
    q = np.true_divide(ratio * 100, np.float32(100))
    q = np.asanyarray(q)
    virt = np.asanyarray((n - 1) * q)
    prev = np.floor(virt)
    k = int(prev)
    gamma = np.asanyarray(virt - prev.astype(np.intp), dtype=virt.dtype)
    if k >= n - 1:
        return max(n - 1, 0), 0.0
    return k, float(gamma)


def _offsets(counts, device):
    off = np.zeros(len(counts) + 1, dtype=np.int32)
    np.cumsum(np.asarray(counts, dtype=np.int64), out=off[1:])
    return torch.from_numpy(off).to(device)


# ----------------------------------------------------------------------------------------------
# a1. sampler
# ----------------------------------------------------------------------------------------------
_OFF_CACHE = {}


def _uniform_offsets(count, B, device):
    key = (int(count), int(B), str(device))
    if key not in _OFF_CACHE:
        _OFF_CACHE[key] = _offsets([count] * B, device)
    return _OFF_CACHE[key]


def sample_tuples(num_points, num_tuples, k, seed, scene_ids=(0,), device=None):
    """Replaces np.random.randint(0, N, (T, k)) (eval.py:207).  Returns int32 [B*T, k] for the scenes in
    scene_ids (an arithmetic progression: base + b*stride)."""
    device = device or _dev()
    B = len(scene_ids)
    stride = (scene_ids[1] - scene_ids[0]) if B > 1 else 1
    assert all(scene_ids[b] == scene_ids[0] + b * stride for b in range(B)), "scene_ids must be arithmetic"
    pt_off = _uniform_offsets(num_points, B, device)
    tup_off = _uniform_offsets(num_tuples, B, device)
    out = torch.empty((B * num_tuples, k), dtype=torch.int32, device=device)
    _lib.check(_L.cppf_sample_tuples(B, _p(pt_off), _p(tup_off), num_tuples, k, C.c_uint64(seed),
                                     int(scene_ids[0]), int(stride), _p(out), _stream()), "cppf_sample_tuples")
    return out


def philox_uniform(num_rows, m, seed, stream_id, scene_ids=(0,), device=None):
    device = device or _dev()
    B = len(scene_ids)
    stride = (scene_ids[1] - scene_ids[0]) if B > 1 else 1
    tup_off = _uniform_offsets(num_rows, B, device)
    out = torch.empty((B * num_rows, m), dtype=torch.float32, device=device)
    _lib.check(_L.cppf_philox_uniform(B, _p(tup_off), num_rows, m, C.c_uint64(seed), int(scene_ids[0]), int(stride),
                                      int(stream_id), _p(out), _stream()), "cppf_philox_uniform")
    return out


# ----------------------------------------------------------------------------------------------
# a3. encode
# ----------------------------------------------------------------------------------------------
def encode_tuples_shot(points, point_idxs_all, shot_feat, normal, pt_off=None, tup_off=None):
    """BeyondCPPF.prepare_tuple_inputs of the SHOT model (train_shot.py:75-83): [T, C(k,2)*4 + k*F]."""
    dev = _dev()
    pts = _t(points, torch.float32, dev)
    idx = _t(point_idxs_all, torch.int32, dev)
    half = isinstance(shot_feat, torch.Tensor) and shot_feat.dtype == torch.float16
    feat = _t(shot_feat, torch.float16 if half else torch.float32, dev)
    nrm = _t(normal, torch.float32, dev)
    T, k = idx.shape
    F = feat.shape[1]
    B = 1 if pt_off is None else pt_off.numel() - 1
    if pt_off is None:
        pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    out = torch.empty((T, k * (k - 1) // 2 * 4 + k * F), dtype=torch.float32, device=dev)
    if half:      # float16 feature table (BASELINE config 5): same layout, float32 rows out
        _lib.check(_L.cppf_encode_tuples_shot_f16(B, _p(pts), _p(nrm), _p(feat), F, _p(idx), k, _p(pt_off), _p(tup_off),
                                                  T, _p(out), _stream()), "cppf_encode_tuples_shot_f16")
        return out
    _lib.check(_L.cppf_encode_tuples_shot(B, _p(pts), _p(nrm), _p(feat), F, _p(idx), k, _p(pt_off), _p(tup_off), T,
                                          _p(out), _stream()), "cppf_encode_tuples_shot")
    return out


def reslayer128_(x, w1, b1, w2):
    """In place x <- x + relu(x w1^T + b1) w2^T for x float32 [rows,128] (device, contiguous), w1 / w2 [128,128] as
    nn.Linear stores them ([out,in]), b1 [128]: the 128-wide identity-skip ResLayer of the tuple / point encoders
    (train_shot.py:19-45) as one matrix-core kernel (cppf_reslayer128).  Returns x."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous()
    w1 = w1.contiguous(); w2 = w2.contiguous(); b1 = b1.contiguous()
    assert w1.shape == (128, 128) and w2.shape == (128, 128) and b1.shape == (128,)
    _lib.check(_L.cppf_reslayer128(_p(x), x.shape[0], _p(w1), _p(b1), _p(w2), _stream()), "cppf_reslayer128")
    return x


def reslayer_tail(x, w1, b1, w0, b0, w2, out=None, scatter_rows=None, valid_count=None, per_group=0):
    """skip(x) + relu(x W1^T + b1) W2^T for a ResLayer with at most 8 outputs (the scale head's ResLayer(64, 3)) as one plain
    float32 kernel (cppf_reslayer_tail).  w1 / w0 [n_out, k_in], w2 [n_out, n_out] as nn.Linear stores them; b0 carries fc2's
    bias.  scatter_rows int32 [rows]: row i is written to out[scatter_rows[i]]; valid_count int32 [rows / per_group]: only
    entries (i % per_group) < valid_count[i // per_group] are written (the padded kept-pair lists of
    VotingPipeline.kept_rows32)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows = x.shape[0]
    n_out, k_in = w1.shape
    w1, w2, b1 = w1.contiguous(), w2.contiguous(), b1.contiguous()
    w0 = None if w0 is None else w0.contiguous()
    b0 = None if b0 is None else b0.contiguous()
    if out is None:
        assert scatter_rows is None, "scatter needs the destination buffer"
        out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[1] >= n_out
    if scatter_rows is not None:
        assert scatter_rows.dtype == torch.int32 and scatter_rows.is_contiguous() and scatter_rows.numel() == rows
    if valid_count is not None:
        assert valid_count.dtype == torch.int32 and valid_count.is_contiguous() and valid_count.numel() * per_group == rows
    _lib.check(_L.cppf_reslayer_tail(_p(x), x.stride(0), int(k_in), int(n_out), rows, _p(w1), _p(b1), _p(w0), _p(b0), _p(w2),
                                     _p(scatter_rows), _p(valid_count), int(per_group), _p(out), out.stride(0), _stream()),
               "cppf_reslayer_tail")
    return out


def nan_to_zero_(x):
    """x[isnan(x)] = 0 in place (eval.py:215-216) as a library kernel; x float32, contiguous, on the device."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    _lib.check(_L.cppf_nan_to_zero(_p(x), x.numel(), _stream()), "cppf_nan_to_zero")
    return x


def cast_f16(x, out=None):
    """float16 copy of a float32 device tensor (round to nearest even) as a library kernel: BASELINE config 5's feature table."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device) if out is None else out
    assert out.dtype == torch.float16 and out.is_contiguous() and out.numel() == x.numel()
    _lib.check(_L.cppf_cast_f16(_p(x), _p(out), x.numel(), _stream()), "cppf_cast_f16")
    return out


def reslayer_split_supported(k_in, n_out, proj, chain=0):
    """True when cppf_reslayer_split has a kernel for a ResLayer of these dims (k_in = columns of x it reads) followed by
    `chain` identity layers of the same width."""
    return k_in % 8 == 0 and _L.cppf_reslayer_split_stream_bytes(int(k_in), int(n_out), int(bool(proj)), int(chain)) > 0


def reslayer_split(x, wq, b1, b0, n_out, out=None, chain=0, tap=None):
    """out = skip(x) + relu(x W1^T + b1) W2^T, then `chain` identity ResLayers of the same width on the result, on the
    bf16 matrix cores in split-float32 arithmetic (cppf_reslayer_split: exact bf16 triples, error within 3 x a float32 GEMM's).  x float32 [rows, k_in] (device; row
    stride a multiple of 4 elements), wq the packed split weight stream (models.pack_split), b1 [(1 + chain) * n_out] (the
    layers' first biases), b0 [n_out] or None for an identity skip (then out defaults to x: in place).  Returns out.
    tap: float32 [rows, n_out] buffer that also receives the activation after the FIRST layer (cppf_reslayer_split_tap): the
    chain behind it continues in registers, its input is never re-read."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, k_in = x.shape
    if out is None:
        out = x if b0 is None else torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (rows, n_out) and out.stride(1) == 1
    b1 = b1.contiguous()
    assert b1.numel() == (1 + chain) * n_out
    b0 = None if b0 is None else b0.contiguous()
    if tap is not None:
        assert tap.dtype == torch.float32 and tap.shape == (rows, n_out) and tap.stride(1) == 1
        _lib.check(_L.cppf_reslayer_split_tap(_p(x), x.stride(0), k_in, _p(tap), tap.stride(0), _p(out), out.stride(0), n_out,
                                              rows, _p(wq), wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
                   "cppf_reslayer_split_tap")
        return out
    _lib.check(_L.cppf_reslayer_split(_p(x), x.stride(0), k_in, _p(out), out.stride(0), n_out, rows, _p(wq),
                                      wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
               "cppf_reslayer_split")
    return out


def reslayer_split16_supported(k_in, n_out, proj, chain=0):
    return k_in % 8 == 0 and _L.cppf_reslayer_split16_stream_bytes(int(k_in), int(n_out), int(bool(proj)), int(chain)) > 0


def reslayer_split16(x, wq, b1, b0, n_out, scale, out=None, chain=0, tap=None, gather=None, decode=None):
    """The ResLayer launches of reslayer_split / _gather / _decode in f16x2 arithmetic (cppf_reslayer_split16; not the default):
    wq = models.pack_split(..., arith="f16x2", scale=scale), b1 / b0 = scale x the biases (scale a power of two).
    gather = (gidx int32 [rows, k], table float32 [points, F]): x are the head columns; decode = (uniforms, prior | None, bins)."""
    from ._lib import ReslayerSplit16Args
    a = ReslayerSplit16Args()
    b1 = b1.contiguous()
    b0 = None if b0 is None else b0.contiguous()
    keep = [b1, b0]
    if gather is not None:
        gidx, table = gather
        rows = gidx.shape[0]
        heads = x
        a.x = heads.data_ptr() if heads.shape[1] else table.data_ptr()
        a.ldx = heads.stride(0) if heads.shape[1] else 0
        a.k_in = heads.shape[1]
        a.gidx, a.slots, a.table, a.fdim = gidx.data_ptr(), gidx.shape[1], table.data_ptr(), table.shape[1]
        out = torch.empty((rows, n_out), dtype=torch.float32, device=table.device)
    else:
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        rows = x.shape[0]
        a.x, a.ldx, a.k_in = x.data_ptr(), x.stride(0), x.shape[1]
    if decode is not None:
        uniforms, prior, bins = decode
        uniforms = uniforms.contiguous()
        if isinstance(prior, BinPrior):
            a.prior_pos, a.prior_inv_sigma = prior.pos.data_ptr(), prior.inv_sigma
            keep.append(prior.pos)
            prior = None
        prior = None if prior is None else prior.contiguous()
        if bins is None:
            bins = torch.empty((rows, 6), dtype=torch.int32, device=x.device)
        a.uniforms, a.bins = uniforms.data_ptr(), bins.data_ptr()
        a.logit_prior = None if prior is None else prior.data_ptr()
        keep += [uniforms, prior]
        n_out = 192
    else:
        if out is None:
            out = x if b0 is None else torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
        a.out, a.ldo = out.data_ptr(), out.stride(0)
    if tap is not None:
        a.first_out, a.ld_first = tap.data_ptr(), tap.stride(0)
    a.n_out, a.rows, a.chain = int(n_out), int(rows), int(chain)
    a.wq, a.wq_bytes = wq.data_ptr(), wq.numel() * wq.element_size()
    a.b1 = b1.data_ptr()
    a.b0 = None if b0 is None else b0.data_ptr()
    a.weight_scale = float(scale)
    a.stream = torch.cuda.current_stream().cuda_stream
    a.sched = _sched().value
    _lib.check(_L.cppf_reslayer_split16(C.byref(a)), "cppf_reslayer_split16")
    return bins if decode is not None else out


class BinPrior:
    """A logit prior given by its generator instead of its [T, 6, 32] array: prior[t, c, k] = -0.5 ((k - pos[t, c]) * inv_sigma)^2,
    a Gaussian bump in logit space around a per-coordinate bin position (the synthetic teacher of the benchmarks and of
    eval.main's synthetic mode; a coarse pose hypothesis would supply the same).  The fused bin draw evaluates it in its epilogue
    (cppf_reslayer_split_decode_prior: prior_pos / prior_inv_sigma; the reference has no prior -- cppf_hip_experimental.h) -- 24 bytes per tuple read instead of 768; dense() is the array, built
    with the same three float32 operations in the same order, so both forms draw the same bins bit for bit."""

    def __init__(self, pos, inv_sigma):
        assert pos.dtype == torch.float32 and pos.dim() == 2 and pos.shape[1] == 6
        self.pos = pos.contiguous()
        self.inv_sigma = float(np.float32(inv_sigma))
        assert self.inv_sigma > 0.0

    def dense(self, nb=32):
        k = torch.arange(nb, device=self.pos.device, dtype=torch.float32)
        z = (k[None, None, :] - self.pos[..., None]) * torch.tensor(self.inv_sigma, dtype=torch.float32, device=self.pos.device)
        return ((z * z) * -0.5).contiguous()


def reslayer_split_decode(x, wq, b1, b0, uniforms, prior=None, bins=None):
    """The logit head's output layer (192-wide projection ResLayer) with the bin draw of eval.py:225-229 as its epilogue
    (cppf_reslayer_split_decode): bins int32 [rows, 6] = what decode_bins draws from these logits (+ prior) at `uniforms`
    [rows, 6], bit for bit; the logits are never written.  prior: None, a float32 [rows, 6, 32] array, or a BinPrior (generated
    in the epilogue)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows = x.shape[0]
    u = uniforms.contiguous()
    assert u.dtype == torch.float32 and u.numel() == rows * 6
    ppos, pis = None, 0.0
    if isinstance(prior, BinPrior):
        assert prior.pos.shape[0] == rows
        ppos, pis, prior = prior.pos, prior.inv_sigma, None
    if prior is not None:
        prior = prior.contiguous()
        assert prior.dtype == torch.float32 and prior.numel() == rows * 192
    if bins is None:
        bins = torch.empty((rows, 6), dtype=torch.int32, device=x.device)
    assert bins.dtype == torch.int32 and bins.is_contiguous() and bins.numel() == rows * 6
    if prior is None and ppos is None:       # the reference's draw (eval.py:225-229): the stable entry point
        _lib.check(_L.cppf_reslayer_split_decode(_p(x), x.stride(0), x.shape[1], rows, _p(wq), wq.numel() * wq.element_size(),
                                                 _p(b1.contiguous()), _p(b0.contiguous()), _p(u), _p(bins), _sched(), _stream()),
                   "cppf_reslayer_split_decode")
        return bins
    _lib.check(_L.cppf_reslayer_split_decode_prior(_p(x), x.stride(0), x.shape[1], rows, _p(wq), wq.numel() * wq.element_size(),
                                                   _p(b1.contiguous()), _p(b0.contiguous()), _p(prior), _p(ppos), C.c_float(pis), _p(u),
                                                   _p(bins), _sched(), _stream()),
               "cppf_reslayer_split_decode_prior")
    return bins


def encode_tuples_shot_heads(points, point_idxs_all, normal, pt_off=None, tup_off=None):
    """The pair-feature block of prepare_tuple_inputs (train_shot.py:75-81: the first 4 C(k,2) columns of
    encode_tuples_shot's rows, bit for bit) and the tuples' global point indices: (heads float32 [T, 4 C(k,2)], gidx int32
    [T, k]) -- the inputs of reslayer_split_gather, which reads the per-point descriptors itself."""
    dev = _dev()
    pts = _t(points, torch.float32, dev)
    nrm = _t(normal, torch.float32, dev)
    idx = _t(point_idxs_all, torch.int32, dev)
    T, k = idx.shape
    B = 1 if pt_off is None else pt_off.numel() - 1
    if pt_off is None:
        pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    ncol = k * (k - 1) // 2 * 4
    heads = torch.empty((T, ncol), dtype=torch.float32, device=dev)
    gidx = torch.empty((T, k), dtype=torch.int32, device=dev)
    _lib.check(_L.cppf_encode_tuples_shot_heads(B, _p(pts), _p(nrm), _p(idx), k, _p(pt_off), _p(tup_off), T, _p(heads), ncol,
                                                _p(gidx), _stream()), "cppf_encode_tuples_shot_heads")
    return heads, gidx


def reslayer_split_gather(heads, gidx, table, wq, b1, b0, n_out, chain=0):
    """The first ResLayer of the tuple encoder (+ `chain` identity layers) on rows [heads | table[gidx[:, 0]] | ... ] that
    are gathered by the kernel instead of being materialised (cppf_reslayer_split_gather); same arithmetic and results as
    reslayer_split on encode_tuples_shot's rows.  Returns float32 [T, n_out]."""
    assert heads.is_cuda and heads.dtype == torch.float32 and heads.dim() == 2
    assert gidx.dtype == torch.int32 and gidx.is_contiguous() and table.dtype == torch.float32 and table.is_contiguous()
    rows = gidx.shape[0]
    out = torch.empty((rows, n_out), dtype=torch.float32, device=heads.device)
    b1 = b1.contiguous()
    b0 = b0.contiguous()
    assert b1.numel() == (1 + chain) * n_out
    ld_heads = heads.stride(0) if heads.shape[1] else 0
    _lib.check(_L.cppf_reslayer_split_gather(_p(heads if heads.shape[1] else table), ld_heads, heads.shape[1], _p(gidx), gidx.shape[1], _p(table),
                                             table.shape[1], _p(out), out.stride(0), n_out, rows, _p(wq),
                                             wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
               "cppf_reslayer_split_gather")
    return out


class TupleSource:
    """What prepare_tuple_inputs reads besides the descriptors: points, the sampler's tuple indices (scene-local with pt_off /
    tup_off: a batch; global without) and -- the SHOT model, train_shot.py:75-83 -- normals (normal=None: the DINO model's
    coordinate block, train_dino.py:92).  Handed to fused_stack(gather=(TupleSource, None, table[, fold])) the first
    tuple-encoder launch builds the head columns itself (reslayer_split_encode: 40 pair features; reslayer_split_sumencode: 30
    coordinate differences); heads() materialises them through the separate kernel for the launch forms that read an array
    (f16x2 arithmetic, the SHOT model's table-fed variant)."""

    def __init__(self, points, point_idxs_all, normal=None, pt_off=None, tup_off=None):
        dev = _dev()
        self.pts = _t(points, torch.float32, dev)
        self.nrm = None if normal is None else _t(normal, torch.float32, dev)
        self.idx = _t(point_idxs_all, torch.int32, dev)
        T, self.k = self.idx.shape
        if pt_off is None:
            pt_off, tup_off = _offsets([self.pts.shape[0]], dev), _offsets([T], dev)
        self.pt_off, self.tup_off = pt_off, tup_off
        self.B = pt_off.numel() - 1
        npair = self.k * (self.k - 1) // 2
        self.shape = (T, npair * 4 if normal is not None else (npair * 3 + 7) // 8 * 8)      # of the head block it stands for

    def heads(self):
        if self.nrm is None:
            return encode_tuples_coord_heads(self.pts, self.idx, self.pt_off, self.tup_off)
        return encode_tuples_shot_heads(self.pts, self.idx, self.nrm, self.pt_off, self.tup_off)


CUS_PER_SHADER_ENGINE = 8         # gfx950: 8 XCDs x 4 shader engines x 8 CUs


def batch_mode_reserved_cus(dev=None):
    """What the two-stream batch mode leaves to the other stream: one CU per shader engine (32 on an MI355X).  A queue's
    workgroups go round-robin over the shader engines, so a single engine without a free CU stalls the other stream's launch
    until the persistent MLP launch ends (docs/measurements.md 11.7)."""
    return torch.cuda.get_device_properties(dev if dev is not None else _dev()).multi_processor_count // CUS_PER_SHADER_ENGINE


def mlp_reserve_cus(cus):
    """cppf_mlp_reserve_cus: every later cppf_reslayer_split* launch of this process leaves `cus` CUs to the kernels of other
    streams (0 = one persistent workgroup per CU).  Results do not depend on it.  Returns the PREVIOUS reservation: nested users
    (eval.run_ensemble inside a BatchMode block, ...) restore that instead of zeroing the outer one; cus < 0 only queries."""
    prev = int(_L.cppf_mlp_reserve_cus(int(cus)))
    if prev < 0:
        _lib.check(prev, "cppf_mlp_reserve_cus")
    return prev


class mlp_cus_reserved:
    """`with ops.mlp_cus_reserved():` -- the batch mode's setting for the launches enqueued inside the block (two streams working
    on different batches: one batch's voting / descriptor kernels run on the reserved CUs beside the other's matrix-core kernels)."""

    def __init__(self, cus=None):
        self.cus = batch_mode_reserved_cus() if cus is None else int(cus)

    def __enter__(self):
        self.prev = mlp_reserve_cus(self.cus)
        return self

    def __exit__(self, *exc):
        mlp_reserve_cus(self.prev)
        return False


def reslayer_split_encode_supported(k, n_out):
    return int(k) == 5 and int(n_out) == 128


def reslayer_split_encode(src, table, wq, b1, b0, n_out, chain=0):
    """reslayer_split_gather with the pair features built inside the kernel (cppf_reslayer_split_encode): rows
    [40 pair features of src | table[idx[:, 0]] | ...], bit-identical to the two-kernel form.  Returns float32 [T, n_out]."""
    assert isinstance(src, TupleSource) and table.dtype == torch.float32 and table.is_contiguous() and table.is_cuda
    rows = src.idx.shape[0]
    out = torch.empty((rows, n_out), dtype=torch.float32, device=table.device)
    b1 = b1.contiguous()
    b0 = b0.contiguous()
    assert b1.numel() == (1 + chain) * n_out
    _lib.check(_L.cppf_reslayer_split_encode(src.B, _p(src.pts), _p(src.nrm), _p(src.idx), src.k, _p(src.pt_off), _p(src.tup_off),
                                             _p(table), table.shape[1], _p(out), out.stride(0), n_out, rows, _p(wq),
                                             wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
               "cppf_reslayer_split_encode")
    return out


def reslayer_split_sumencode(src, tables, wq, b1, b0, n_out, chain=0):
    """reslayer_split_sumgather with the coordinate columns built inside the kernel (cppf_reslayer_split_sumencode): bit-identical to
    the two-kernel form.  src: TupleSource without normals; tables float32 [points, slots * 256].  Returns float32 [T, n_out]."""
    assert isinstance(src, TupleSource) and src.nrm is None and tables.dtype == torch.float32 and tables.stride(1) == 1 and tables.is_cuda
    rows = src.idx.shape[0]
    assert tables.shape[1] == src.k * 256
    out = torch.empty((rows, n_out), dtype=torch.float32, device=tables.device)
    b1 = b1.contiguous()
    b0 = b0.contiguous()
    assert b1.numel() == (1 + chain) * n_out
    _lib.check(_L.cppf_reslayer_split_sumencode(src.B, _p(src.pts), _p(src.idx), src.k, _p(src.pt_off), _p(src.tup_off), _p(tables),
                                                tables.stride(0), _p(out), out.stride(0), n_out, rows, _p(wq),
                                                wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
               "cppf_reslayer_split_sumencode")
    return out


def linear_split_supported(k_in, n_out):
    """True when cppf_linear_split evaluates an nn.Linear of these dims (input columns a multiple of 8, outputs of 256)."""
    return k_in % 8 == 0 and _L.cppf_linear_split_stream_bytes(int(k_in), int(n_out)) > 0


def linear_split(x, wq, bias, n_out, out=None, scale=None):
    """out = x W^T + bias, a plain nn.Linear on the matrix cores in the same split-float32 arithmetic (cppf_linear_split; the
    per-point transforms of the DINO model, train_dino.py:86-87).  x float32 [rows, k_in] (device; row stride a multiple of 4),
    wq = models.pack_linear(W, k_in), bias float32 [n_out] or None, n_out a multiple of 256.
    scale: f16x2 arithmetic (cppf_reslayer_split16, mode 1) with wq / bias packed at that weight scale."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, k_in = x.shape
    if out is None:
        out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (rows, n_out) and out.stride(1) == 1
    bias = None if bias is None else bias.contiguous()
    if scale is not None:
        from ._lib import ReslayerSplit16Args
        a = ReslayerSplit16Args()
        a.x, a.ldx, a.k_in = x.data_ptr(), x.stride(0), k_in
        a.out, a.ldo, a.n_out, a.rows = out.data_ptr(), out.stride(0), int(n_out), int(rows)
        a.wq, a.wq_bytes = wq.data_ptr(), wq.numel() * wq.element_size()
        a.b1 = None if bias is None else bias.data_ptr()
        a.weight_scale, a.mode = float(scale), 1
        a.stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_L.cppf_reslayer_split16(C.byref(a)), "cppf_reslayer_split16(linear)")
        return out
    _lib.check(_L.cppf_linear_split(_p(x), x.stride(0), k_in, _p(out), out.stride(0), int(n_out), rows, _p(wq),
                                    wq.numel() * wq.element_size(), _p(bias), _stream()), "cppf_linear_split")
    return out


def encode_tuples_coord_heads(points, point_idxs_all, pt_off=None, tup_off=None):
    """The coordinate block of the DINO model's prepare_tuple_inputs (train_dino.py:92: encode_tuples_coord's columns, bit for
    bit) zero-padded to a multiple of 8 columns, and the tuples' global point indices: (heads float32 [T, round8(3 C(k,2))],
    gidx int32 [T, k]) -- the inputs of reslayer_split_sumgather."""
    dev = _dev()
    pts = _t(points, torch.float32, dev)
    idx = _t(point_idxs_all, torch.int32, dev)
    T, k = idx.shape
    B = 1 if pt_off is None else pt_off.numel() - 1
    if pt_off is None:
        pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    ncol = (k * (k - 1) // 2 * 3 + 7) // 8 * 8
    heads = torch.empty((T, ncol), dtype=torch.float32, device=dev)
    gidx = torch.empty((T, k), dtype=torch.int32, device=dev)
    _lib.check(_L.cppf_encode_tuples_coord_heads(B, _p(pts), _p(idx), k, _p(pt_off), _p(tup_off), T, _p(heads), ncol, _p(gidx),
                                                 _stream()), "cppf_encode_tuples_coord_heads")
    return heads, gidx


def reslayer_split_sumgather(heads, gidx, tables, wq, b1, b0, n_out, chain=0, scale=None):
    """The first ResLayer of a tuple encoder (+ `chain` identity layers) on rows [heads | s] whose per-point part s enters through
    per-point slot tables (cppf_reslayer_split_sumgather): tables float32 [points, slots * 256] = [W1_i p | W0_i p] per point and
    slot, summed into the accumulators in slot order; only the head columns go through the first product.  Returns [T, n_out].
    scale: f16x2 arithmetic (cppf_reslayer_split16, mode 2)."""
    assert heads.is_cuda and heads.dtype == torch.float32 and heads.dim() == 2 and heads.stride(1) == 1
    assert gidx.dtype == torch.int32 and gidx.is_contiguous() and tables.dtype == torch.float32 and tables.stride(1) == 1
    rows, slots = gidx.shape
    assert tables.shape[1] == slots * 256
    out = torch.empty((rows, n_out), dtype=torch.float32, device=heads.device)
    b1 = b1.contiguous()
    b0 = b0.contiguous()
    assert b1.numel() == (1 + chain) * n_out
    if scale is not None:
        from ._lib import ReslayerSplit16Args
        a = ReslayerSplit16Args()
        a.x, a.ldx, a.k_in = heads.data_ptr(), heads.stride(0), heads.shape[1]
        a.out, a.ldo, a.n_out, a.rows, a.chain = out.data_ptr(), out.stride(0), int(n_out), int(rows), int(chain)
        a.wq, a.wq_bytes = wq.data_ptr(), wq.numel() * wq.element_size()
        a.b1, a.b0 = b1.data_ptr(), b0.data_ptr()
        a.gidx, a.slots, a.table, a.ld_table = gidx.data_ptr(), int(slots), tables.data_ptr(), tables.stride(0)
        a.weight_scale, a.mode = float(scale), 2
        a.stream = torch.cuda.current_stream().cuda_stream
        a.sched = _sched().value
        _lib.check(_L.cppf_reslayer_split16(C.byref(a)), "cppf_reslayer_split16(sumgather)")
        return out
    _lib.check(_L.cppf_reslayer_split_sumgather(_p(heads), heads.stride(0), heads.shape[1], _p(gidx), slots, _p(tables),
                                                tables.stride(0), _p(out), out.stride(0), n_out, rows, _p(wq),
                                                wq.numel() * wq.element_size(), _p(b1), _p(b0), int(chain), _sched(), _stream()),
               "cppf_reslayer_split_sumgather")
    return out


def encode_tuples_coord(points, point_idxs_all, out=None, pt_off=None, tup_off=None):
    """Coordinate part of the DINO model's prepare_tuple_inputs (train_dino.py:92): [T, C(k,2)*3]
    (written into the leading columns of `out` if given)."""
    dev = _dev()
    pts = _t(points, torch.float32, dev)
    idx = _t(point_idxs_all, torch.int32, dev)
    T, k = idx.shape
    B = 1 if pt_off is None else pt_off.numel() - 1
    if pt_off is None:
        pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    ncol = k * (k - 1) // 2 * 3
    if out is None:
        out = torch.empty((T, ncol), dtype=torch.float32, device=dev)
    assert out.is_contiguous() and out.shape[0] == T and out.shape[1] >= ncol and out.dtype == torch.float32
    _lib.check(_L.cppf_encode_tuples_coord(B, _p(pts), _p(idx), k, _p(pt_off), _p(tup_off), T, _p(out),
                                           out.shape[1], _stream()), "cppf_encode_tuples_coord")
    return out


# ----------------------------------------------------------------------------------------------
# a4 + a5. decode
# ----------------------------------------------------------------------------------------------
def encode_tuples_dino(tables, bias, point_idxs_all, out, out_col, pt_off=None, tup_off=None):
    """Descriptor part of the DINO model's prepare_tuple_inputs (train_dino.py:95-96) as a gather-add of per-point
    products: out[:, out_col:out_col+D] = bias + sum_i tables[idx[:, i], i, :] (tables [N, k, D], see cppf_hip.h)."""
    dev = _dev()
    tb = _t(tables, torch.float32, dev)
    n, k, D = tb.shape
    idx = _t(point_idxs_all, torch.int32, dev)
    T = idx.shape[0]
    if pt_off is None:
        pt_off, tup_off = _offsets([n], dev), _offsets([T], dev)
    B = pt_off.numel() - 1
    bs = None if bias is None else _t(bias, torch.float32, dev)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[0] != T:
        raise CppfError("encode_tuples_dino: out must be a contiguous float32 [T, C] tensor")
    _lib.check(_L.cppf_encode_tuples_dino(B, _p(tb), k, D, _p(bs), _p(idx), _p(pt_off), _p(tup_off), T, _p(out),
                                          out.shape[1], int(out_col), _stream()), "cppf_encode_tuples_dino")
    return out


def decode_bins(pred_cls, uniforms, points, point_idxs_all, up, right, front, pt_off=None, tup_off=None, prior=None):
    """eval.py:225-240.  pred_cls [T,6,nb] logits, uniforms [T,6] in [0,1).  (up, right, front) are the three
    axis vectors in the positional order of generate_target_pairs' signature (the reference call site passes
    cfg.up, cfg.front, cfg.right).  Returns dict(bins, pred_pairs_scaled [T,2,3], scale, targets_tr, targets_rot)."""
    dev = _dev()
    lg = _t(pred_cls, torch.float32, dev)
    T, six, nb = lg.shape
    assert six == 6
    un = _t(uniforms, torch.float32, dev).reshape(T, 6)
    pts = _t(points, torch.float32, dev)
    idx = _t(point_idxs_all, torch.int32, dev)
    k = idx.shape[1]
    B = 1 if pt_off is None else pt_off.numel() - 1
    if pt_off is None:
        pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    bins = torch.empty((T, 6), dtype=torch.int32, device=dev)
    scaled = torch.empty((T, 2, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((T,), dtype=torch.float32, device=dev)
    tr = torch.empty((T, 2), dtype=torch.float32, device=dev)
    rot = torch.empty((T, 3), dtype=torch.float32, device=dev)
    pr = None if prior is None else _t(prior, torch.float32, dev).reshape(T, 6, nb)
    if pr is None:
        _lib.check(_L.cppf_decode_bins(B, _p(lg), nb, _p(un), _p(pts), _p(idx), k, _p(pt_off), _p(tup_off), T,
                                       _axes9(up, right, front), _p(bins), _p(scaled), _p(scale), _p(tr), _p(rot),
                                       _stream()), "cppf_decode_bins")
    else:
        _lib.check(_L.cppf_decode_bins_prior(B, _p(lg), _p(pr), nb, _p(un), _p(pts), _p(idx), k, _p(pt_off), _p(tup_off), T,
                                             _axes9(up, right, front), _p(bins), _p(scaled), _p(scale), _p(tr), _p(rot),
                                             _stream()), "cppf_decode_bins_prior")
    return dict(bins=bins, pred_pairs_scaled=scaled, scale=scale, targets_tr=tr, targets_rot=rot)


def generate_target_pairs(point_pairs, up, right, front, center=np.zeros((3,))):
    """dataset.py:118-135: NumPy in, NumPy out (float32 [T,2], float32 [T,3]), computed on the GPU."""
    dev = _dev()
    pp = _t(point_pairs, torch.float32, dev).reshape(-1, 6)
    T = pp.shape[0]
    tup_off = _offsets([T], dev)
    ctr = _t(np.asarray(center, dtype=np.float64).reshape(1, 3), torch.float64, dev)
    tr = torch.empty((T, 2), dtype=torch.float32, device=dev)
    rot = torch.empty((T, 3), dtype=torch.float32, device=dev)
    _lib.check(_L.cppf_generate_target_pairs(1, _p(pp), _p(tup_off), T, _axes9(up, right, front), _p(ctr), _p(tr),
                                             _p(rot), _stream()), "cppf_generate_target_pairs")
    return tr.cpu().numpy(), rot.cpu().numpy()


# ----------------------------------------------------------------------------------------------
# a6. vote_center
# ----------------------------------------------------------------------------------------------
def scene_bounds(pts, pt_off, res):
    B = pt_off.numel() - 1
    grids = torch.empty((B, 32), dtype=torch.uint8, device=pts.device)
    _lib.check(_L.cppf_scene_bounds(B, _p(pts), _p(pt_off), C.c_float(res), _p(grids), _stream()), "cppf_scene_bounds")
    return grids


def grids_to_host(grids):
    raw = grids.cpu().numpy().tobytes()
    n = len(raw) // 32
    return (SceneGrid * n).from_buffer_copy(raw)


def vote_center(pc, preds_tr, res, point_idxs, num_rots=36, vis=None, trig=None, mode=0, return_device=False,
                weights=None):
    """train_dino.py:171-215.  Returns (grid_obj int64 ndarray [gx,gy,gz], cand_world float64 ndarray [3]).
    weights (extension, BASELINE config 5): per-pair vote weights in [0, 4], accumulated as round(w*256)."""
    dev = _dev()
    pts = _t(pc, torch.float32, dev)
    tr = _t(preds_tr, torch.float32, dev)
    idx = _t(point_idxs, torch.int32, dev)
    T, k = idx.shape
    cs, sn = _trig(num_rots, trig, dev)
    pt_off, tup_off = _offsets([pts.shape[0]], dev), _offsets([T], dev)
    grids = scene_bounds(pts, pt_off, res)
    g = grids_to_host(grids)[0]                       # the reference syncs here as well (.cpu() at :206)
    if g.flags & 2 or g.ncell <= 0:
        raise CppfError("vote_center: grid of %s cells is not representable" % (list(g.g),))
    G = int(g.ncell)
    grid = torch.empty((G,), dtype=torch.int32, device=dev)
    grid_off = torch.tensor([0, G], dtype=torch.int64, device=dev)
    ws_bytes = _L.cppf_vote_center_workspace_bytes(1, G, T)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    argmax = torch.empty((1,), dtype=torch.int64, device=dev)
    peak = torch.empty((1,), dtype=torch.int32, device=dev)
    world = torch.empty((1, 3), dtype=torch.float64, device=dev)
    wt = None if weights is None else _t(weights, torch.float32, dev).reshape(-1)
    _lib.check(_L.cppf_vote_center(1, _p(pts), _p(pt_off), _p(idx), k, _p(tup_off), T, T, _p(tr), _p(wt), C.c_double(res),
                                   num_rots, _p(cs), _p(sn), _p(grids), _p(grid), _p(grid_off), G, mode, _p(ws),
                                   ws_bytes, _p(argmax), _p(peak), _p(world), _stream()), "cppf_vote_center")
    shape = (int(g.g[0]), int(g.g[1]), int(g.g[2]))
    if return_device:
        return grid.view(shape), world[0], argmax, peak
    grid_obj = grid.to(torch.int64).cpu().numpy().reshape(shape)
    return grid_obj, world[0].cpu().numpy()


# ----------------------------------------------------------------------------------------------
# a8 / a9 with the reference's signatures
# ----------------------------------------------------------------------------------------------
def vote_rotation(pc, preds_rot, point_idxs, num_rots=36, trig=None):
    """train_dino.py:218-239.  Returns (up float32 tensor [n, num_rots, 3], mask bool tensor [T])."""
    dev = _dev()
    pts = _t(pc, torch.float32, dev)
    ang = _t(preds_rot, torch.float32, dev)
    idx = _t(point_idxs, torch.int32, dev)
    T, k = idx.shape
    cs, sn = _trig(num_rots, trig, dev)
    up = torch.empty((max(T, 1), num_rots, 3), dtype=torch.float32, device=dev)
    valid = torch.empty((max(T, 1),), dtype=torch.uint8, device=dev)
    nvalid = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((max(T, 1) * 4,), dtype=torch.uint8, device=dev)
    _lib.check(_L.cppf_vote_rotation(_p(pts), pts.shape[0], _p(idx), k, T, _p(ang), num_rots, _p(cs), _p(sn), _p(up),
                                     _p(valid), _p(nvalid), _p(ws), ws.numel(), _stream()), "cppf_vote_rotation")
    n = int(nvalid.item())
    return up[:n], valid[:T].bool()


def sphere_counts(pred, sphere_pts, bmm_size, angle_tol, wt=None):
    """The accumulation half of get_topk_dir (eval.py:41-45): float32 counts [S] on the device."""
    dev = _dev()
    cand = _t(pred, torch.float32, dev).reshape(-1, 3)
    sph = _t(sphere_pts, torch.float32, dev)
    M, S = cand.shape[0], sph.shape[0]
    w = None if wt is None else _t(wt, torch.float64, dev).reshape(-1)
    if w is not None:
        assert w.numel() == M
    nch = max((M + bmm_size - 1) // bmm_size, 1)
    ws = torch.empty((nch * S * 8,), dtype=torch.uint8, device=dev)
    counts = torch.empty((S,), dtype=torch.float32, device=dev)
    _lib.check(_L.cppf_sphere_counts(_p(cand), M, _p(w), _p(sph), S, C.c_float(cone_threshold(angle_tol)),
                                     int(bmm_size), _p(counts), _p(ws), ws.numel(), _stream()), "cppf_sphere_counts")
    return counts


def get_topk_dir(pred, sphere_pts, bmm_size, angle_tol, wt=None, topk=1):
    """eval.py:37-51.  Returns (dirs float32 ndarray [topk,3], counts float32 ndarray [topk])."""
    counts = sphere_counts(pred, sphere_pts, bmm_size, angle_tol, wt)
    topk_idx = torch.topk(counts, topk)[1].cpu().numpy()
    sphere_np = np.asarray(sphere_pts)
    return np.array(sphere_np[topk_idx]), counts.cpu().numpy()[topk_idx]


# ----------------------------------------------------------------------------------------------
# steps in front of the path
# ----------------------------------------------------------------------------------------------
def backproject(depth, intrinsics, instance_mask, return_device=False):
    """Masked depth map -> float32 camera-frame points on the GPU: utils/util.py:2586-2607 followed by the sign flip
    and float32 cast every caller applies (eval.py:185-189), i.e. the cloud eval.py actually uses.  Pixel order is
    np.where's (row-major).  Returns (pc float32[n,3], (rows, cols)); NumPy arrays unless return_device."""
    dev = _dev()
    d = _t(depth, torch.float32, dev)
    m = _t(np.asarray(instance_mask) != 0 if not isinstance(instance_mask, torch.Tensor) else instance_mask != 0,
           torch.uint8, dev)
    if d.dim() != 2 or d.shape != m.shape:
        raise CppfError("backproject: depth and mask must be 2-D and of equal shape")
    H, W = d.shape
    kinv = (C.c_double * 9)(*np.linalg.inv(np.asarray(intrinsics, dtype=np.float64).reshape(3, 3)).reshape(9))
    cap = H * W
    pts = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    rc = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_L.cppf_backproject(_p(d), _p(m), H, W, kinv, cap, _p(pts), _p(rc), _p(cnt), _stream()),
               "cppf_backproject")
    n = int(cnt.item())
    pts, rc = pts[:n], rc[:n]
    if return_device:
        return pts, (rc[:, 0], rc[:, 1])
    rc = rc.cpu().numpy().astype(np.int64)
    return pts.cpu().numpy(), (rc[:, 0], rc[:, 1])


def backproject_reference(depth, intrinsics, instance_mask):
    """utils/util.py:2586-2607 as the reference returns it: (pts float64[n,3] with x and y negated, (rows, cols) int64) --
    cppf_backproject64: float64 depth in, the reference's float64 operations in its order, bit-identical to NumPy's array."""
    dev = _dev()
    d = _t(depth, torch.float64, dev)
    m = _t(np.asarray(instance_mask) != 0 if not isinstance(instance_mask, torch.Tensor) else instance_mask != 0,
           torch.uint8, dev)
    if d.dim() != 2 or d.shape != m.shape:
        raise CppfError("backproject: depth and mask must be 2-D and of equal shape")
    H, W = d.shape
    kinv = (C.c_double * 9)(*np.linalg.inv(np.asarray(intrinsics, dtype=np.float64).reshape(3, 3)).reshape(9))
    cap = H * W
    pts = torch.empty((cap, 3), dtype=torch.float64, device=dev)
    rc = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_L.cppf_backproject64(_p(d), _p(m), H, W, kinv, cap, _p(pts), _p(rc), _p(cnt), _stream()),
               "cppf_backproject64")
    n = int(cnt.item())
    rc = rc[:n].cpu().numpy().astype(np.int64)
    return pts[:n].cpu().numpy(), (rc[:, 0], rc[:, 1])


def downsample(pc, res, seed=0, return_device=False):
    """Indices of one uniformly random point per `res` voxel (utils/util.py:39-46), ascending.  The draw is
    Philox(seed, point index) instead of NumPy's global RandomState, so it is reproducible on any device."""
    dev = _dev()
    pts = _t(pc, torch.float32, dev).reshape(-1, 3)
    n = pts.shape[0]
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = _L.cppf_voxel_downsample_workspace_bytes(n)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    _lib.check(_L.cppf_voxel_downsample(_p(pts), n, float(res), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(idx), _p(cnt),
                                        _p(ws), wsb, _stream()), "cppf_voxel_downsample")
    idx = idx[: int(cnt.item())].long()
    return idx if return_device else idx.cpu().numpy()


def interpolate_features(descriptors, pts, strides=8, normalize=True, half=False):
    """interpolate_features(descriptors[1,C,h,w], pts[1,n,2]) (dataset.py:40-59) on the GPU.  `descriptors` may be any
    strided view of the token map (e.g. tokens.reshape(1,h,w,C).permute(0,3,1,2), as dataset.py:78 builds it): the
    kernel reads it in place.  Returns [1, C, n] like the reference (a transposed view of the keypoint-major result;
    `.squeeze(0).T` is contiguous); half=True gives float16."""
    dev = _dev()
    d = descriptors if isinstance(descriptors, torch.Tensor) else torch.as_tensor(np.asarray(descriptors))
    d = d.to(device=dev, dtype=torch.float32)
    if d.dim() == 3:
        d = d[None]
    if d.dim() != 4 or d.shape[0] != 1:
        raise CppfError("interpolate_features: descriptors must be [1, C, h, w]")
    _, Cc, h, w = d.shape
    p = _t(pts, torch.float32, dev).reshape(-1, 2)
    n = p.shape[0]
    out = torch.empty((n, Cc), dtype=torch.float16 if half else torch.float32, device=dev)
    sc, sy, sx = d.stride(1), d.stride(2), d.stride(3)
    _lib.check(_L.cppf_interpolate_features(_p(d), Cc, h, w, sc, sy, sx, _p(p), n, C.c_float(strides),
                                            int(bool(normalize)), _p(out), int(bool(half)), _stream()),
               "cppf_interpolate_features")
    return out.T[None]
