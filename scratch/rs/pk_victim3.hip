// The packed float32 forms the SLP vectoriser emitted in rot_bins_lut_kernel (op_sel / op_sel_hi / neg modifiers) against the
// scalar instructions, solo and while a 448-register MFMA wavefront of another stream shares the SIMD.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 pk(float lo, float hi) { return (u64)__float_as_uint(lo) | ((u64)__float_as_uint(hi) << 32); }
__device__ __forceinline__ bool same(u64 v, float lo, float hi) { return (unsigned)v == __float_as_uint(lo) && (unsigned)(v >> 32) == __float_as_uint(hi); }
#define NPAT 8
__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, int iters) {
  extern __shared__ char smem[];
  float a = 1.0f + (float)threadIdx.x * 0.00390625f + (float)(blockIdx.x & 1023) * 1e-4f, b = 0.5f + (float)threadIdx.x * 0.001f;
  unsigned nb[NPAT] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    const float x0 = a, x1 = b, y0 = b + 2.0f, y1 = a * 0.75f;
    const u64 x = pk(x0, x1), y = pk(y0, y1);
    u64 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    nb[0] += same(r, x1 + y0, x0 + y1) ? 0u : 1u;
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(x));
    nb[1] += same(r, x0 + x1, x1 + x0) ? 0u : 1u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    nb[2] += same(r, x0 * y1, x1 * y1) ? 0u : 1u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    nb[3] += same(r, x0 * y0, x0 * y1) ? 0u : 1u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r) : "v"(x), "v"(y));
    nb[4] += same(r, x0 * y0, x1 * y0) ? 0u : 1u;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    nb[5] += same(r, x0 - y0, x1 - y1) ? 0u : 1u;
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(r) : "v"(x), "v"(y));
    nb[6] += same(r, x1, y0) ? 0u : 1u;
    u64 m;
    asm volatile("v_pk_mul_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %1, %2, %3, %0" : "=&v"(m), "=&v"(r) : "v"(x), "v"(y));
    nb[7] += same(r, fmaf(x0, y0, -(x0 * y0)), fmaf(x1, y1, -(x1 * y1))) ? 0u : 1u;
    a = a * 1.0009765625f + 0.0625f;
    a = (a > 1000.0f) ? a * 0.0009765625f : a;
    b = b * 0.99951171875f + 0.03125f;
  }
#pragma unroll
  for (int p = 0; p < NPAT; ++p) if (nb[p]) atomicAdd(&bad[p], nb[p]);
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
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(d, 0, 64)); CK(hipDeviceSynchronize());
      if (with_mlp)
        for (int r = 0; r < 2; ++r) {
          int rc = cppf_reslayer_split(x, 256, 256, x, 256, 256, rows, wq, bytes, b, nullptr, 0, sa);
          if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
        }
      hipLaunchKernelGGL(victim_kernel, dim3(8192), dim3(256), 34000, sb, d, 3000);
      CK(hipDeviceSynchronize());
      unsigned h[NPAT];
      CK(hipMemcpy(h, d, 4 * NPAT, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: wrong of %lld each:", with_mlp ? "beside " : "without", 8192ll * 256 * 3000);
      for (int p = 0; p < NPAT; ++p) printf(" %u", h[p]);
      printf("\n");
    }
  return 0;
}
