// Does float arithmetic (correctly rounded sqrt / division sequences, conversions, LDS float4 reads) of a small co-resident
// wavefront stay deterministic while a 448-register MLP wavefront runs on the same SIMD?  Each thread evaluates the same
// chain twice per iteration and compares the two results bit for bit.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

__device__ __noinline__ float chain(float a, float b, float c, const float4* tab, int k) {
  const float4 q = tab[k & 1023];
  const float n = fmaxf(__builtin_sqrtf(fmaf(c, c, fmaf(b, b, a * a))), 1e-7f);
  const float x = a / n, y = b / n, z = c / n;
  const float d = fmaf(z, q.z, fmaf(y, q.y, x * q.x));
  const float t = (x * 3.25f + y) / (fabsf(z) + 0.5f);
  const int ci = (int)((1.0f - y) * 128.0f);
  return d + t + (float)ci + __builtin_sqrtf(fabsf(t));
}

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* info, int iters) {
  __shared__ float4 tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const unsigned h = hashu(i * 31u + 7u);
    tab[i] = make_float4((float)(h & 1023) / 1024.f - 0.5f, (float)((h >> 10) & 1023) / 1024.f - 0.5f, (float)(h >> 20) / 4096.f - 0.5f, 0.f);
  }
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    const unsigned h = hashu(blockIdx.x * 7919u + threadIdx.x * 131u + i);
    const float a = (float)(h & 1023) / 512.f - 1.0f, b = (float)((h >> 10) & 1023) / 512.f - 1.0f, c = (float)(h >> 20) / 2048.f - 1.0f;
    const float r1 = chain(a, b, c, tab, (int)h);
    __builtin_amdgcn_s_sleep(2);
    float a2 = a, b2 = b, c2 = c;
    asm volatile("" : "+v"(a2), "+v"(b2), "+v"(c2));
    const float r2 = chain(a2, b2, c2, tab, (int)h);
    if (__float_as_uint(r1) != __float_as_uint(r2)) {
      if (atomicAdd(bad, 1u) == 0) { info[0] = blockIdx.x; info[1] = threadIdx.x; info[2] = i; info[3] = __float_as_uint(r1); info[4] = __float_as_uint(r2); }
    }
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
      hipLaunchKernelGGL(victim_kernel, dim3(4096), dim3(256), 0, sb, d, d + 1, 2000);
      CK(hipDeviceSynchronize());
      unsigned h[6];
      CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: %u of %d evaluations differ between two back-to-back runs of the same arithmetic", with_mlp ? "beside " : "without", h[0], 4096 * 256 * 2000);
      if (h[0]) printf(" (first: block %u thread %u iter %u %08x vs %08x)", h[1], h[2], h[3], h[4], h[5]);
      printf("\n");
    }
  return 0;
}
