"""ctypes loader of oracle/libvote_oracle.so (C + OpenMP restatement of get_topk_dir's sphere-bin count, eval.py:37-51).
TEST INFRASTRUCTURE, NOT PRODUCT: the NumPy function oracle.cppf_oracle.get_topk_dir is the definition, this is the same
arithmetic at the speed a CPU baseline deserves (tests/test_oracle_golden.py: bit-equal counts, any thread count)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvote_oracle.so")
_lib = None


def build():
    src = os.path.join(_HERE, "vote_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libvote_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.vote_oracle_sphere_counts.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_long,
                                                   C.c_int, C.c_void_p]
        _lib.vote_oracle_sphere_counts.restype = None
        _lib.vote_oracle_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(_load().vote_oracle_max_threads())


def sphere_counts(pred, sphere_pts, wt, thr, bmm_size, threads=0):
    """counts f32[S] of get_topk_dir: pred f32[M,3], sphere_pts f32[S,3], wt f64[M] or [M,1] (divisors), thr float32."""
    pred = np.ascontiguousarray(pred, dtype=np.float32).reshape(-1, 3)
    sph = np.ascontiguousarray(sphere_pts, dtype=np.float32).reshape(-1, 3)
    wt = np.ascontiguousarray(np.asarray(wt, dtype=np.float64).reshape(-1))
    assert wt.shape[0] == pred.shape[0]
    counts = np.zeros((sph.shape[0],), np.float32)
    _load().vote_oracle_sphere_counts(pred.ctypes.data, pred.shape[0], sph.ctypes.data, sph.shape[0], wt.ctypes.data,
                                      C.c_float(float(thr)), int(bmm_size), int(threads), counts.ctypes.data)
    return counts
