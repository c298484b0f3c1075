"""Builds libcppf_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and `python -m cppf2_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcppf_hip.so")
SOURCES = ["cppf_core.hip", "cppf_vote.hip", "cppf_shot.hip", "cppf_prep.hip", "cppf_refine.hip", "cppf_mlp.hip", "cppf_mlp_split.hip"]
# -ffp-contract=off: every float op rounds where it is written (bit-exact vote grid); fused ops are explicit fmaf().
# -munsafe-fp-atomics: native ds/global float64 atomic add for the rotation-bin partial sums.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-munsafe-fp-atomics", "-fPIC", "-Wall", "-Wno-unused-function"]


# -fno-slp-vectorize (every file): no compiler-generated packed float32 arithmetic.  Two reasons, both measured in round 3:
# (1) gfx950 erratum (scratch/rs/pk_victim4.hip, profiles/r3_pk_op_sel_erratum.md): v_pk_{mul,add,fma}_f32 whose LOW lane takes the
#     HIGH half of source 1 (op_sel:[x,1]) occasionally computes that lane with source 1 = 0 while a wavefront of the wide MLP
#     kernels (another workgroup, another stream) runs on the same SIMD (1e-8 .. 1e-7 of the evaluations; never alone).  The SLP vectoriser emits exactly those forms
#     (37 of them in the vote / SHOT / refinement kernels) -- harmless while one stream runs one kernel at a time, wrong votes as
#     soon as a second stream runs the MLP kernels beside them.  tests/test_abi.py disassembles the built library and fails on any
#     such instruction.
# (2) a packed float32 instruction between two MFMAs holds the matrix pipe for 16 cycles (scratch/rs/filler_price.hip: 48.8
#     instead of 32.2 cycles per MFMA), which is why cppf_mlp_split.hip was built this way first.
FLAGS.append("-fno-slp-vectorize")
FILE_FLAGS = {}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    inc = os.path.join(ROOT, "include")
    hdrs = [os.path.join(inc, "cppf_hip.h"), os.path.join(inc, "cppf_hip_experimental.h"), os.path.join(CSRC, "cppf_common.h")]
    extra = os.environ.get("CPPF_EXTRA_FLAGS", "").split()
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + extra + ["-I", inc, "-I", CSRC, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
