"""ctypes loader of oracle/libshot_oracle.so (C restatement of PCL 1.9.1 normals + SHOT352).
TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (see shot_oracle.c header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libshot_oracle.so")


def build():
    src = os.path.join(_HERE, "shot_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libshot_oracle.so"])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.shot_oracle_compute.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.shot_oracle_compute.restype = None
        _lib.shot_oracle_compute_ex.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
        _lib.shot_oracle_compute_ex.restype = None
        _lib.shot_oracle_compute_color.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
        _lib.shot_oracle_compute_color.restype = None
        _lib.shot_oracle_normals.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        _lib.shot_oracle_normals.restype = None
        _lib.shot_oracle_set_threads.argtypes = [C.c_int]
        _lib.shot_oracle_set_threads.restype = None
        _lib.shot_oracle_max_threads.restype = C.c_int
    return _lib


def compute(pc, normal_r, shot_r):
    """Returns (shot f32[N,352], normal f32[N,3], rf f32[N,9])."""
    pc = np.ascontiguousarray(pc, dtype=np.float32).reshape(-1, 3)
    n = pc.shape[0]
    shot = np.empty((n, 352), np.float32)
    normal = np.empty((n, 3), np.float32)
    rf = np.empty((n, 9), np.float32)
    _load().shot_oracle_compute(pc.ctypes.data, n, C.c_float(normal_r), C.c_float(shot_r), shot.ctypes.data,
                                normal.ctypes.data, rf.ctypes.data)
    return shot, normal, rf


def normals(pc, normal_r):
    pc = np.ascontiguousarray(pc, dtype=np.float32).reshape(-1, 3)
    out = np.empty((pc.shape[0], 3), np.float32)
    _load().shot_oracle_normals(pc.ctypes.data, pc.shape[0], C.c_float(normal_r), out.ctypes.data)
    return out


def max_threads():
    return int(_load().shot_oracle_max_threads())


def compute_ex(pc, normal_r, shot_r, pcl_arithmetic=False, threads=1):
    """threads: 1 = single-threaded like PCL's estimators (src_shot/shot.cpp:66-89; the default), 0 = all cores (OpenMP over the
    query points; identical outputs).
    Like compute(), plus diagnostics and the PCL-arithmetic mode (see shot_oracle.c).  Returns (shot, normal, rf,
    diag f64[N,9]): LRF eigenvalues (descending), the sign tallies 2*plus - n of the x and z axes, the smallest
    margin of a neighbour to a decision boundary of PCL's interpolation (cosine step, radial shell, equator, octant), the histogram's L2 norm, the neighbour count,
    the smallest |d^2 - r^2| / r^2 over all points (proximity of a point to the rim of the support)."""
    pc = np.ascontiguousarray(pc, dtype=np.float32).reshape(-1, 3)
    n = pc.shape[0]
    shot = np.empty((n, 352), np.float32)
    normal = np.empty((n, 3), np.float32)
    rf = np.empty((n, 9), np.float32)
    diag = np.empty((n, 9), np.float64)
    _load().shot_oracle_set_threads(int(threads))
    _load().shot_oracle_compute_ex(pc.ctypes.data, n, C.c_float(normal_r), C.c_float(shot_r), int(bool(pcl_arithmetic)),
                                   shot.ctypes.data, normal.ctypes.data, rf.ctypes.data, diag.ctypes.data)
    return shot, normal, rf, diag


def compute_color(pc, pc_color, normal_r, shot_r):
    """shot.compute_color (src_shot/shot.cpp:102-161; SHOT1344 = 352 shape + 992 colour entries).
    Returns (shot f32[N,1344], normal f32[N,3], diag f64[N,9])."""
    pc = np.ascontiguousarray(pc, dtype=np.float32).reshape(-1, 3)
    col = np.ascontiguousarray(pc_color, dtype=np.float32).reshape(-1, 3)
    n = pc.shape[0]
    assert col.shape[0] == n
    shot = np.empty((n, 1344), np.float32)
    normal = np.empty((n, 3), np.float32)
    diag = np.empty((n, 9), np.float64)
    _load().shot_oracle_compute_color(pc.ctypes.data, col.ctypes.data, n, C.c_float(normal_r), C.c_float(shot_r),
                                      shot.ctypes.data, normal.ctypes.data, diag.ctypes.data)
    return shot, normal, diag
