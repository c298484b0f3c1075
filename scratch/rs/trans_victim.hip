// Is the result of a transcendental instruction (v_rcp_f32 / v_sqrt_f32: the quarter-rate unit) visible to a dependent VALU
// instruction issued at the compiler's minimum distance (one wait state, or two independent VALU instructions) while a
// 448-register MFMA wavefront of another stream shares the SIMD?  Each thread evaluates the same operation at the minimum
// distance and with 16 idle states in between and compares the two results bit for bit.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* info, int iters) {
  extern __shared__ char smem[];
  float a = 1.0f + (float)threadIdx.x * 0.00390625f + (float)(blockIdx.x & 1023) * 1e-4f;
  unsigned nb0 = 0, nb1 = 0, nb2 = 0;
  for (int i = 0; i < iters; ++i) {
    float r, p, r2, p2, t0, t1;
    // pattern 0: rcp, one wait state, dependent multiply
    asm volatile("v_rcp_f32 %0, %2\n s_nop 0\n v_mul_f32 %1, %0, %2" : "=&v"(r), "=&v"(p) : "v"(a));
    asm volatile("v_rcp_f32 %0, %2\n s_nop 7\n s_nop 7\n v_mul_f32 %1, %0, %2" : "=&v"(r2), "=&v"(p2) : "v"(a));
    nb0 += (__float_as_uint(p) != __float_as_uint(p2)) ? 1u : 0u;
    // pattern 1: rcp, two independent VALU instructions, dependent fma (the division expansion's schedule)
    asm volatile("v_rcp_f32 %0, %4\n v_add_f32 %2, %4, %4\n v_mul_f32 %3, %4, %4\n v_fma_f32 %1, -%4, %0, 1.0" : "=&v"(r), "=&v"(p), "=&v"(t0), "=&v"(t1) : "v"(a));
    asm volatile("v_rcp_f32 %0, %4\n v_add_f32 %2, %4, %4\n v_mul_f32 %3, %4, %4\n s_nop 7\n s_nop 7\n v_fma_f32 %1, -%4, %0, 1.0" : "=&v"(r2), "=&v"(p2), "=&v"(t0), "=&v"(t1) : "v"(a));
    nb1 += (__float_as_uint(p) != __float_as_uint(p2)) ? 1u : 0u;
    // pattern 2: sqrt, one wait state, dependent integer add (the sqrt expansion's schedule)
    asm volatile("v_sqrt_f32 %0, %2\n s_nop 0\n v_add_u32 %1, -1, %0" : "=&v"(r), "=&v"(p) : "v"(a));
    asm volatile("v_sqrt_f32 %0, %2\n s_nop 7\n s_nop 7\n v_add_u32 %1, -1, %0" : "=&v"(r2), "=&v"(p2) : "v"(a));
    nb2 += (__float_as_uint(p) != __float_as_uint(p2)) ? 1u : 0u;
    a = a * 1.0009765625f + 0.0625f;
    a = (a > 1000.0f) ? a * 0.0009765625f : a;
  }
  if (nb0 | nb1 | nb2) {
    atomicAdd(&bad[0], nb0); atomicAdd(&bad[1], nb1); atomicAdd(&bad[2], nb2);
    info[0] = blockIdx.x; info[1] = threadIdx.x;
  }
}

int main() {
  const int64_t rows = 400000;
  float *x, *b;
  CK(hipMalloc(&x, rows * 256 * 4));
  CK(hipMalloc(&b, 16 * 256 * 4));
  CK(hipMemset(x, 0, rows * 256 * 4));
  CK(hipMemset(b, 0, 16 * 256 * 4));
  unsigned* d;
  CK(hipMalloc(&d, 64));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  const int64_t bytes = cppf_reslayer_split_stream_bytes(256, 256, 0, 0);
  void* wq;
  CK(hipMalloc(&wq, bytes));
  CK(hipMemset(wq, 0x3c, bytes));
  for (int with_mlp = 0; with_mlp < 2; ++with_mlp)
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(d, 0, 64));
      CK(hipDeviceSynchronize());
      if (with_mlp)
        for (int r = 0; r < 2; ++r) {
          int rc = cppf_reslayer_split(x, 256, 256, x, 256, 256, rows, wq, bytes, b, nullptr, 0, sa);
          if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
        }
      hipLaunchKernelGGL(victim_kernel, dim3(8192), dim3(256), 34000, sb, d, d + 4, 4000);
      CK(hipDeviceSynchronize());
      unsigned h[6];
      CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: of %lld evaluations each, differ from the long-distance result: rcp + 1 wait state %u, rcp + 2 independent VALU %u, sqrt + 1 wait state %u\n",
             with_mlp ? "beside " : "without", 8192ll * 256 * 4000, h[0], h[1], h[2]);
    }
  return 0;
}
