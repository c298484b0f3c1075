// Do packed float32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: SLP-vectorised code) of a small wavefront give the
// same results as the scalar instructions while a 448-register MFMA wavefront of another stream shares the SIMD?
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
__device__ __forceinline__ u64 pk(float lo, float hi) { return (u64)__float_as_uint(lo) | ((u64)__float_as_uint(hi) << 32); }
__device__ __forceinline__ float lo32(u64 v) { return __uint_as_float((unsigned)v); }
__device__ __forceinline__ float hi32(u64 v) { return __uint_as_float((unsigned)(v >> 32)); }

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* info, int iters) {
  extern __shared__ char smem[];
  float a = 1.0f + (float)threadIdx.x * 0.00390625f + (float)(blockIdx.x & 1023) * 1e-4f, b = 0.5f + (float)threadIdx.x * 0.001f;
  unsigned nb0 = 0, nb1 = 0, nb2 = 0;
  for (int i = 0; i < iters; ++i) {
    const float x0 = a, x1 = b, y0 = b, y1 = a * 0.75f;
    const u64 x = pk(x0, x1), y = pk(y0, y1);
    u64 pm, pa, pf;
    float m0, m1, s0, s1, f0, f1;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(pm) : "v"(x), "v"(y));
    asm volatile("v_mul_f32 %0, %2, %3\n v_mul_f32 %1, %4, %5" : "=&v"(m0), "=&v"(m1) : "v"(x0), "v"(y0), "v"(x1), "v"(y1));
    nb0 += (__float_as_uint(lo32(pm)) != __float_as_uint(m0) || __float_as_uint(hi32(pm)) != __float_as_uint(m1)) ? 1u : 0u;
    // packed op followed at once by a consumer of its result (the compiler puts one wait state there)
    u64 ps;
    asm volatile("v_pk_mul_f32 %0, %2, %3\n s_nop 0\n v_pk_add_f32 %1, %0, %0" : "=&v"(pa), "=&v"(ps) : "v"(x), "v"(y));
    s0 = m0 + m0; s1 = m1 + m1;
    nb1 += (__float_as_uint(lo32(ps)) != __float_as_uint(s0) || __float_as_uint(hi32(ps)) != __float_as_uint(s1)) ? 1u : 0u;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(pf) : "v"(x), "v"(y), "v"(pm));
    asm volatile("v_fma_f32 %0, %2, %3, %4\n v_fma_f32 %1, %5, %6, %7" : "=&v"(f0), "=&v"(f1) : "v"(x0), "v"(y0), "v"(m0), "v"(x1), "v"(y1), "v"(m1));
    nb2 += (__float_as_uint(lo32(pf)) != __float_as_uint(f0) || __float_as_uint(hi32(pf)) != __float_as_uint(f1)) ? 1u : 0u;
    a = a * 1.0009765625f + 0.0625f;
    a = (a > 1000.0f) ? a * 0.0009765625f : a;
    b = b * 0.99951171875f + 0.03125f;
  }
  if (nb0 | nb1 | nb2) { atomicAdd(&bad[0], nb0); atomicAdd(&bad[1], nb1); atomicAdd(&bad[2], nb2); info[0] = blockIdx.x; info[1] = threadIdx.x; }
}

int main() {
  const int64_t rows = 400000;
  float *x, *b;
  CK(hipMalloc(&x, rows * 256 * 4)); CK(hipMalloc(&b, 16 * 256 * 4));
  CK(hipMemset(x, 0, rows * 256 * 4)); CK(hipMemset(b, 0, 16 * 256 * 4));
  unsigned* d;
  CK(hipMalloc(&d, 64));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  const int64_t bytes = cppf_reslayer_split_stream_bytes(256, 256, 0, 0);
  void* wq;
  CK(hipMalloc(&wq, bytes)); CK(hipMemset(wq, 0x3c, bytes));
  for (int with_mlp = 0; with_mlp < 2; ++with_mlp)
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(d, 0, 64)); CK(hipDeviceSynchronize());
      if (with_mlp)
        for (int r = 0; r < 2; ++r) {
          int rc = cppf_reslayer_split(x, 256, 256, x, 256, 256, rows, wq, bytes, b, nullptr, 0, sa);
          if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
        }
      hipLaunchKernelGGL(victim_kernel, dim3(8192), dim3(256), 34000, sb, d, d + 4, 4000);
      CK(hipDeviceSynchronize());
      unsigned h[6];
      CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: of %lld evaluations each, packed != scalar: pk_mul %u, pk_mul + dependent add %u, pk_fma %u\n",
             with_mlp ? "beside " : "without", 8192ll * 256 * 4000, h[0], h[1], h[2]);
    }
  return 0;
}
