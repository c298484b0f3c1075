"""Drop-in for the reference's native module `src_shot.build.shot` (src_shot/shot.cpp:164-168).

    from cppf2_amd import shot
    shot_feat, normal = shot.compute(pc, normal_r, shot_r)     # eval.py:210

Same call signature and return convention as the pybind11 module: NumPy float arrays in (float64 is
silently converted like pybind11's forcecast, shot.cpp:45), a list of two 1-D float32 arrays out that
the caller reshapes to [N,352] / [N,3] (eval.py:211,214); rows PCL leaves undefined are NaN.
The computation runs on the GPU (cppf_shot352 in libcppf_hip.so); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops

_L = _lib.load()

# Arithmetic of the normals (include/cppf_hip.h, flags of the SHOT entry points): "pcl" (default) = pcl::NormalEstimation's -- what
# src_shot/shot.cpp:25-32, 66-72 runs: single-pass float32 sums of the raw coordinates in (distance, index) order, closed-form
# eigen33, float32 viewpoint flip; "f64" = float64 covariance about the query point + Jacobi (rounds 1-3; more accurate, but not
# what the reference's checkpoints were trained on).  Every function below takes arithmetic=None = this module default.
ARITHMETIC = "pcl"


def _flags(arithmetic):
    a = ARITHMETIC if arithmetic is None else arithmetic
    if a not in ("pcl", "f64"):
        raise ValueError("arithmetic must be 'pcl' or 'f64', not %r" % (a,))
    return 1 if a == "f64" else 0          # CPPF_SHOT_F64_NORMALS


_WS = {}          # (device, stream) -> scratch buffer
_PREPARED = {}    # (device, stream) -> (B, n, pts pointer) of the last prepare_device, checked by describe_device


WS_CACHE_MAX = 8        # (device, stream) pairs whose scratch buffer stays alive; the least recently used one goes first


def _key(dev):
    dev = torch.device(dev)
    index = dev.index if dev.index is not None else torch.cuda.current_device()     # "cuda" and "cuda:0" are one device
    return (int(index), int(torch.cuda.current_stream(dev).cuda_stream))


def _workspace(B, n, dev):
    """One cached scratch buffer per (device index, stream): grown on demand and reused by every call issued on that stream,
    so calls on different streams (or threads using their own streams) never share scratch memory.  At most WS_CACHE_MAX
    buffers are kept (least recently used evicted: a process that keeps creating streams does not pin a workspace per
    stream for its lifetime)."""
    need = _L.cppf_shot352_workspace_bytes(B, n)
    key = _key(dev)
    ws = _WS.pop(key, None)
    if ws is None or ws.numel() < need:
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _PREPARED.pop(key, None)          # a new buffer holds no prepared state
    _WS[key] = ws                         # (re-)inserted last: dicts keep insertion order, the first key is the oldest
    while len(_WS) > WS_CACHE_MAX:
        old = next(iter(_WS))
        del _WS[old]
        _PREPARED.pop(old, None)
    return ws


def compute_device(pts, pt_off, normal_r, shot_r, want_rf=False, arithmetic=None):
    """Batched device API: pts float32 [Ntot,3] (device), pt_off int32 [B+1] (device).
    Returns (shot [Ntot,352], normal [Ntot,3][, rf [Ntot,9]]) device tensors."""
    dev = pts.device
    n = pts.shape[0]
    B = pt_off.numel() - 1
    out_shot = torch.empty((n, 352), dtype=torch.float32, device=dev)
    out_normal = torch.empty((n, 3), dtype=torch.float32, device=dev)
    out_rf = torch.empty((n, 9), dtype=torch.float32, device=dev) if want_rf else None
    ws = _workspace(B, n, dev)
    _PREPARED.pop(_key(dev), None)
    _lib.check(_L.cppf_shot352(B, ops._p(pts), ops._p(pt_off), n, C.c_float(normal_r), C.c_float(shot_r),
                               ops._p(out_shot), ops._p(out_normal), ops._p(out_rf), ops._p(ws), ws.numel(),
                               _flags(arithmetic), ops._stream()), "cppf_shot352")
    if want_rf:
        return out_shot, out_normal, out_rf
    return out_shot, out_normal


def normals_device(pts, pt_off, normal_r, out=None, arithmetic=None):
    n = pts.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if out is None else out
    ws = _workspace(pt_off.numel() - 1, n, pts.device)
    _PREPARED.pop(_key(pts.device), None)
    _lib.check(_L.cppf_estimate_normals(pt_off.numel() - 1, ops._p(pts), ops._p(pt_off), n, C.c_float(normal_r),
                                        ops._p(out), ops._p(ws), ws.numel(), _flags(arithmetic), ops._stream()),
               "cppf_estimate_normals")
    return out


def descriptors_device(pts, pt_off, normals, shot_r, out=None):
    n = pts.shape[0]
    out = torch.empty((n, 352), dtype=torch.float32, device=pts.device) if out is None else out
    ws = _workspace(pt_off.numel() - 1, n, pts.device)
    _PREPARED.pop(_key(pts.device), None)
    _lib.check(_L.cppf_shot352_from_normals(pt_off.numel() - 1, ops._p(pts), ops._p(pt_off), n, ops._p(normals),
                                            C.c_float(shot_r), ops._p(out), None, ops._p(ws), ws.numel(),
                                            ops._stream()), "cppf_shot352_from_normals")
    return out


def prepare_device(pts, pt_off, normal_r, shot_r, out_normal=None, arithmetic=None):
    """First half of compute_device (cell sort, covariances, eigen-solves): returns the normals."""
    n = pts.shape[0]
    B = pt_off.numel() - 1
    out_normal = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if out_normal is None else out_normal
    ws = _workspace(B, n, pts.device)
    _lib.check(_L.cppf_shot_prepare(B, ops._p(pts), ops._p(pt_off), n, C.c_float(normal_r), C.c_float(shot_r),
                                    ops._p(out_normal), ops._p(ws), ws.numel(), _flags(arithmetic), ops._stream()),
               "cppf_shot_prepare")
    _PREPARED[_key(pts.device)] = (B, n, pts.data_ptr(), float(shot_r), ws.data_ptr())
    return out_normal


def describe_device(pts, pt_off, normals, shot_r, out=None, nan_to_zero=False):
    """Second half of compute_device (histogram kernel); must directly follow prepare_device on the same inputs.
    nan_to_zero folds eval.py:215's np.nan_to_num(shot_feat) into the kernel's store."""
    n = pts.shape[0]
    B = pt_off.numel() - 1
    out = torch.empty((n, 352), dtype=torch.float32, device=pts.device) if out is None else out
    ws = _workspace(B, n, pts.device)
    # the workspace carries prepare_device's cell tables, frames and neighbour lists: it must be the same buffer, on the
    # same stream, prepared for the same cloud, with no other SHOT call in between
    if _PREPARED.get(_key(pts.device)) != (B, n, pts.data_ptr(), float(shot_r), ws.data_ptr()):
        raise _lib.CppfError("describe_device must directly follow prepare_device on the same stream with the same "
                             "points, batch layout and descriptor radius")
    _lib.check(_L.cppf_shot_describe(B, ops._p(pts), ops._p(pt_off), n, ops._p(normals), C.c_float(shot_r),
                                     int(bool(nan_to_zero)), ops._p(out), None, ops._p(ws), ws.numel(), ops._stream()), "cppf_shot_describe")
    return out


def compute(pc, normal_r=0.1, shot_r=0.17, arithmetic=None):
    """shot.compute (src_shot/shot.cpp:45-100): returns [float32[N*352], float32[N*3]].  The normals follow
    pcl::NormalEstimation's arithmetic (ARITHMETIC = "pcl") unless arithmetic="f64" asks for the float64 ones."""
    dev = ops._dev()
    pts = ops._t(np.asarray(pc, dtype=np.float32).reshape(-1, 3), torch.float32, dev)
    pt_off = ops._offsets([pts.shape[0]], dev)
    s, n = compute_device(pts, pt_off, float(normal_r), float(shot_r), arithmetic=arithmetic)
    return [s.reshape(-1).cpu().numpy(), n.reshape(-1).cpu().numpy()]


def compute_color_device(pts, colors, pt_off, normal_r, shot_r, arithmetic=None):
    """Batched device API of compute_color: pts / colors float32 [Ntot,3] (device).  Returns (shot1344 [Ntot,1344], normal)."""
    dev = pts.device
    n = pts.shape[0]
    B = pt_off.numel() - 1
    out = torch.empty((n, 1344), dtype=torch.float32, device=dev)
    out_normal = torch.empty((n, 3), dtype=torch.float32, device=dev)
    need = _L.cppf_shot1344_workspace_bytes(B, n)
    ws = torch.empty((max(need, 256),), dtype=torch.uint8, device=dev)
    _lib.check(_L.cppf_shot1344(B, ops._p(pts), ops._p(colors), ops._p(pt_off), n, C.c_float(normal_r), C.c_float(shot_r),
                                ops._p(out), ops._p(out_normal), ops._p(ws), ws.numel(), _flags(arithmetic), ops._stream()),
               "cppf_shot1344")
    return out, out_normal


def compute_color(pc, pc_color, normal_r=0.1, shot_r=0.17, arithmetic=None):
    """shot.compute_color (src_shot/shot.cpp:102-161): float32[N*1344] (SHOT1344: 352 shape + 992 colour entries)."""
    dev = ops._dev()
    pts = ops._t(np.asarray(pc, dtype=np.float32).reshape(-1, 3), torch.float32, dev)
    col = ops._t(np.asarray(pc_color, dtype=np.float32).reshape(-1, 3), torch.float32, dev)
    assert col.shape[0] == pts.shape[0]
    pt_off = ops._offsets([pts.shape[0]], dev)
    s, _ = compute_color_device(pts, col, pt_off, float(normal_r), float(shot_r), arithmetic=arithmetic)
    return s.reshape(-1).cpu().numpy()


def estimate_normal(pc, normal_r, arithmetic=None):
    """shot.estimate_normal (src_shot/shot.cpp:12-42): returns float32[N*3]."""
    dev = ops._dev()
    pts = ops._t(np.asarray(pc, dtype=np.float32).reshape(-1, 3), torch.float32, dev)
    pt_off = ops._offsets([pts.shape[0]], dev)
    return normals_device(pts, pt_off, float(normal_r), arithmetic=arithmetic).reshape(-1).cpu().numpy()
