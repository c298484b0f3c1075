"""The wavefront idioms of cppf_common.h (ballot masks, mbcnt prefix, DPP scan / row shifts, permlane swaps, the butterfly sum, the
correctly rounded square root without the denormal path) against the portable forms they replace, on the GPU: a test-only
translation unit (tests/csrc/wave_idioms_check.hip) is compiled here with the library's flags and driven through ctypes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    from cppf2_amd import build
    out = tmp_path_factory.mktemp("idioms") / "libwave_idioms_check.so"
    flags = [f for f in build.FLAGS if f not in ("-Wall",)]
    subprocess.check_call([hipcc] + flags + ["-shared", "-I", os.path.join(ROOT, "include"), "-I", build.CSRC,
                                             os.path.join(ROOT, "tests", "csrc", "wave_idioms_check.hip"), "-o", str(out)])
    return C.CDLL(str(out))


def test_sqrt_rn_equals_the_correctly_rounded_sqrtf_on_every_float32(lib):
    out = (C.c_ulonglong * 2)()
    assert lib.check_sqrt_all(out) == 0
    assert out[0] == 0, "sqrt_rn differs from sqrtf on %d inputs, e.g. bit pattern 0x%08x" % (out[0], out[1])


def test_wavefront_idioms_equal_their_portable_forms(lib):
    rng = np.random.RandomState(11)
    rows = 4096
    v = rng.randint(0, 2 ** 32, size=(rows, 64), dtype=np.uint64).astype(np.uint32)
    v[0] = 0
    v[1] = 0xffffffff
    v[2, ::2] = 0                                                     # alternating predicate
    d = (rng.randn(rows, 64) * np.exp(rng.randn(rows, 64) * 8)).astype(np.float64)
    d[3] = 0.0
    bad = (C.c_ulonglong * 8)()
    assert lib.check_idioms(v.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(C.POINTER(C.c_double)), rows, bad) == 0
    names = ["wave_ballot", "lanes_below", "wave_inclusive_scan_u32", "upper_half(u32)", "upper_half(f64)", "row_down<1>", "row_down<2>",
             "wave_sum"]
    assert list(bad) == [0] * 8, dict(zip(names, list(bad)))
