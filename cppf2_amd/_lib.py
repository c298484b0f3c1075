"""ctypes binding of libcppf_hip.so (C ABI: include/cppf_hip.h = the stable part, include/cppf_hip_experimental.h = the rest).

There is no CPU fallback: if the library is missing the import of any op fails loudly with
instructions to build it.  The library is kept in-tree (cppf2_amd/libcppf_hip.so).
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcppf_hip.so")
if os.environ.get("CPPF_LIB"):           # a differently built library (probe builds under scratch/); same ABI version required
    LIB_PATH = os.path.abspath(os.environ["CPPF_LIB"])

ABI_VERSION = 11


class CppfError(RuntimeError):
    pass


class SceneGrid(C.Structure):
    _fields_ = [("c0", C.c_float * 3), ("g", C.c_int32 * 3), ("ncell", C.c_int32), ("flags", C.c_int32)]


class SceneResult(C.Structure):
    _fields_ = [("argmax", C.c_int64), ("t", C.c_double * 3), ("R", C.c_double * 9), ("scale", C.c_float * 3),
                ("peak", C.c_uint32), ("up_idx", C.c_int32), ("right_idx", C.c_int32), ("kept", C.c_int32),
                ("up_count", C.c_float), ("right_count", C.c_float), ("flags", C.c_int32), ("ncell", C.c_int32), ("pad_", C.c_int32 * 3)]


class ReslayerSplit16Args(C.Structure):
    """CppfReslayerSplit16Args (include/cppf_hip.h): one struct for every launch form of the f16x2 ResLayer kernels."""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("k_in", C.c_int32),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("n_out", C.c_int32), ("rows", C.c_int64),
                ("wq", C.c_void_p), ("wq_bytes", C.c_int64), ("b1", C.c_void_p), ("b0", C.c_void_p), ("chain", C.c_int32),
                ("weight_scale", C.c_float),
                ("first_out", C.c_void_p), ("ld_first", C.c_int64),
                ("gidx", C.c_void_p), ("slots", C.c_int32), ("table", C.c_void_p), ("fdim", C.c_int32),
                ("logit_prior", C.c_void_p), ("uniforms", C.c_void_p), ("bins", C.c_void_p),
                ("stream", C.c_void_p), ("mode", C.c_int32), ("ld_table", C.c_int64), ("sched", C.c_void_p),
                ("prior_pos", C.c_void_p), ("prior_inv_sigma", C.c_float)]


assert C.sizeof(SceneGrid) == 32
assert C.sizeof(SceneResult) == 160

_p = C.c_void_p
_i = C.c_int
_i32 = C.c_int32
_i64 = C.c_int64
_u64 = C.c_uint64
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes).  STABLE lists every symbol include/cppf_hip.h declares, EXPERIMENTAL every symbol of
# include/cppf_hip_experimental.h (tests/test_abi.py checks both, symbol by symbol).
STABLE = {
    "cppf_version": (_i, []),
    "cppf_last_error_string": (C.c_char_p, []),
    "cppf_sample_tuples": (_i, [_i, _p, _p, _i, _i, _u64, _i32, _i32, _p, _p]),
    "cppf_philox_uniform": (_i, [_i, _p, _i, _i, _u64, _i32, _i32, _i32, _p, _p]),
    "cppf_shot352_workspace_bytes": (_i64, [_i, _i64]),
    "cppf_shot352": (_i, [_i, _p, _p, _i64, _f, _f, _p, _p, _p, _p, _i64, _i, _p]),
    "cppf_shot1344_workspace_bytes": (_i64, [_i, _i64]),
    "cppf_shot1344": (_i, [_i, _p, _p, _p, _i64, _f, _f, _p, _p, _p, _i64, _i, _p]),
    "cppf_shot_prepare": (_i, [_i, _p, _p, _i64, _f, _f, _p, _p, _i64, _i, _p]),
    "cppf_shot_describe": (_i, [_i, _p, _p, _i64, _p, _f, _i, _p, _p, _p, _i64, _p]),
    "cppf_estimate_normals": (_i, [_i, _p, _p, _i64, _f, _p, _p, _i64, _i, _p]),
    "cppf_encode_tuples_shot": (_i, [_i, _p, _p, _p, _i, _p, _i, _p, _p, _i64, _p, _p]),
    "cppf_encode_tuples_coord": (_i, [_i, _p, _p, _i, _p, _p, _i64, _p, _i, _p]),
    "cppf_decode_bins": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "cppf_generate_target_pairs": (_i, [_i, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "cppf_scene_bounds": (_i, [_i, _p, _p, _f, _p, _p]),
    "cppf_vote_center_workspace_bytes": (_i64, [_i, _i64, _i64]),
    "cppf_vote_center": (_i, [_i, _p, _p, _p, _i, _p, _i, _i64, _p, _p, _d, _i, _p, _p, _p, _p, _p, _i64, _i, _p, _i64,
                              _p, _p, _p, _p]),
    "cppf_backvote_workspace_bytes": (_i64, [_i64, _i]),
    "cppf_backvote_filter": (_i, [_i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _d, _i, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _i64, _p]),
    "cppf_rot_bins_workspace_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "cppf_rot_bins": (_i, [_i, _p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _i, _p, _i, _i,
                           _p, _p, _p, _p, _i64, _p]),
    "cppf_rot_bins2": (_i, [_i, _p, _p, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _i, _p, _i, _i,
                            _p, _p, _p, _p, _i64, _p]),
    "cppf_mlp_reserve_cus": (_i, [_i32]),
    "cppf_kept_rows32": (_i, [_i, _p, _p, _p, _i, _p, _p]),
    "cppf_nan_to_zero": (_i, [_p, _i64, _p]),
    "cppf_reslayer_tail": (_i, [_p, _i64, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "cppf_vote_rotation": (_i, [_p, _i, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "cppf_sphere_counts": (_i, [_p, _i64, _p, _p, _i, _f, _i, _p, _p, _i64, _p]),
    "cppf_refine_pose": (_i, [_i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _f, _p, _p]),
    "cppf_interpolate_features": (_i, [_p, _i, _i, _i, _i64, _i64, _i64, _p, _i, _f, _i, _p, _i, _p]),
    "cppf_backproject": (_i, [_p, _p, _i, _i, _p, _i, _p, _p, _p, _p]),
    "cppf_backproject64": (_i, [_p, _p, _i, _i, _p, _i, _p, _p, _p, _p]),
    "cppf_voxel_downsample_workspace_bytes": (_i64, [_i64]),
    "cppf_voxel_downsample": (_i, [_p, _i, _f, _u64, _p, _p, _p, _i64, _p]),
    "cppf_reslayer_split_stream_bytes": (_i64, [_i, _i, _i, _i]),
    "cppf_reslayer_split": (_i, [_p, _i64, _i, _p, _i64, _i, _i64, _p, _i64, _p, _p, _i, _p, _p]),
    "cppf_reslayer_split_tap": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _i, _i64, _p, _i64, _p, _p, _i, _p, _p]),
    "cppf_reslayer_split_decode": (_i, [_p, _i64, _i, _i64, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "cppf_decode_from_bins": (_i, [_i, _p, _i, _p, _p, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "cppf_reslayer_split_sumencode": (_i, [_i, _p, _p, _i, _p, _p, _p, _i64, _p, _i64, _i, _i64, _p, _i64, _p, _p, _i, _p, _p]),
    "cppf_reslayer_split_encode": (_i, [_i, _p, _p, _p, _i, _p, _p, _p, _i, _p, _i64, _i, _i64, _p, _i64, _p, _p, _i, _p, _p]),
    "cppf_linear_split_stream_bytes": (_i64, [_i32, _i32]),
    "cppf_linear_split": (_i, [_p, _i64, _i32, _p, _i64, _i32, _i64, _p, _i64, _p, _p]),
    "cppf_alignment_loss": (_i, [_i, _p, _p, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p]),
    "cppf_ensemble_select": (_i, [_i, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "cppf_assemble_pose": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
}

EXPERIMENTAL = {
    "cppf_shot352_from_normals": (_i, [_i, _p, _p, _i64, _p, _f, _p, _p, _p, _i64, _p]),
    "cppf_cast_f16": (_i, [_p, _p, _i64, _p]),
    "cppf_encode_tuples_shot_f16": (_i, [_i, _p, _p, _p, _i, _p, _i, _p, _p, _i64, _p, _p]),
    "cppf_encode_tuples_dino": (_i, [_i, _p, _i, _i, _p, _p, _p, _p, _i64, _p, _i, _i, _p]),
    "cppf_kept_rows": (_i, [_i, _p, _p, _p, _i, _p, _p]),
    "cppf_reslayer128": (_i, [_p, _i64, _p, _p, _p, _p]),
    "cppf_encode_tuples_shot_heads": (_i, [_i, _p, _p, _p, _i, _p, _p, _i64, _p, _i, _p, _p]),
    "cppf_reslayer_split_gather": (_i, [_p, _i64, _i, _p, _i, _p, _i, _p, _i64, _i, _i64, _p, _i64, _p, _p, _i, _p, _p]),
    "cppf_encode_tuples_coord_heads": (_i, [_i, _p, _p, _i, _p, _p, _i64, _p, _i32, _p, _p]),
    "cppf_reslayer_split_sumgather": (_i, [_p, _i64, _i32, _p, _i32, _p, _i64, _p, _i64, _i32, _i64, _p, _i64, _p, _p, _i32, _p, _p]),
    "cppf_reslayer_split16_stream_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "cppf_reslayer_split16": (_i, [_p]),
    "cppf_reslayer_split_debug_grid": (_i, [_i32]),
    "cppf_decode_bins_prior": (_i, [_i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "cppf_reslayer_split_decode_prior": (_i, [_p, _i64, _i, _i64, _p, _i64, _p, _p, _p, _p, C.c_float, _p, _p, _p, _p]),
}

SIGNATURES = dict(STABLE, **EXPERIMENTAL)

# Call tracing for tests/test_zz_stable_abi_coverage_gpu.py: with CPPF_ABI_TRACE set when the library is loaded, every call through
# the object load() returns records its symbol in CALLED (the product path never sets it: ops call the ctypes functions directly).
CALLED = set()


class _Traced:
    def __init__(self, lib):
        self._lib = lib
        for name in SIGNATURES:
            setattr(self, name, self._wrap(name, getattr(lib, name)))

    @staticmethod
    def _wrap(name, fn):
        def call(*a):
            CALLED.add(name)
            return fn(*a)
        call.__name__ = name
        return call

    def __getattr__(self, name):            # anything else (a missing symbol raises like the CDLL would)
        return getattr(self._lib, name)


_lib = None


def load():
    """Loads the shared library once; raises CppfError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CppfError(
            "libcppf_hip.so not found at %s -- build it with `python -m cppf2_amd.build` "
            "(hipcc --offload-arch=gfx950).  cppf2_amd has no CPU fallback." % LIB_PATH)
    # torch ships its own libamdhip64.so.7; it must be the HIP runtime of the process (it owns the device
    # context of every tensor we are handed), so make sure it is loaded before our NEEDED entry is resolved.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    v = lib.cppf_version()
    if v != ABI_VERSION:
        raise CppfError("libcppf_hip.so ABI version %d != expected %d; rebuild" % (v, ABI_VERSION))
    _lib = _Traced(lib) if os.environ.get("CPPF_ABI_TRACE") else lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().cppf_last_error_string()
        raise CppfError("%s failed with status %d: %s" % (what, status, (msg or b"").decode("utf-8", "replace")))
