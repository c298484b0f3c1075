"""Out-of-place D = C + A B^T through rocblas_gemm_ex (C != D) against torch.addmm's copy + in-place GEMM (not part of
the product): is the 0.5 ms copy of the kept tuple features avoidable with a plain library call?"""
import ctypes as C, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"))
h = C.c_void_p()
assert lib.rocblas_create_handle(C.byref(h)) == 0
lib.rocblas_set_stream.argtypes = [C.c_void_p, C.c_void_p]
assert lib.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
F32, NONE, TRANS = 151, 111, 112
lib.rocblas_gemm_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int32, C.c_uint32]
T, n = 64 * 20000, 256
g = torch.Generator(device="cpu").manual_seed(0)
feat = torch.randn((T, n), generator=g).cuda(); hh = torch.randn((T, n), generator=g).cuda()
w2 = (torch.randn((n, n), generator=g) * 0.05).cuda()
out = torch.empty_like(feat)
one = C.c_float(1.0)

def oop():
    # row-major out[T,n] = feat + hh @ w2^T  ==  col-major out^T (n x T) = w2 (as op(A) = A^T of the [k x m] view) hh^T + feat^T
    rc = lib.rocblas_gemm_ex(h, TRANS, NONE, n, T, n, C.byref(one), w2.data_ptr(), F32, n, hh.data_ptr(), F32, n, C.byref(one),
                             feat.data_ptr(), F32, n, out.data_ptr(), F32, n, F32, 0, 0, 0)
    assert rc == 0, rc

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

with torch.no_grad():
    oop(); torch.cuda.synchronize()
    want = torch.addmm(feat, hh, w2.t())
    print("max diff", (out - want).abs().max().item())
    print("torch.addmm out of place (copy + GEMM) %.3f ms" % timeit(lambda: torch.addmm(feat, hh, w2.t())))
    x = feat.clone()
    print("torch addmm_ in place                  %.3f ms" % timeit(lambda: x.addmm_(hh, w2.t())))
    print("rocblas_gemm_ex C != D                 %.3f ms" % timeit(oop))
