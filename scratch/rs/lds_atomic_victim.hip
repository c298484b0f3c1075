// Do LDS 64-bit atomics / plain LDS read-modify-write of a small kernel stay exact while the 512-register MLP kernel runs on
// another stream?  Each workgroup adds a known multiset into LDS bins and checks every bin against a second, atomic-free count.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
#include <vector>
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* info, int iters) {
  extern __shared__ unsigned long long acc[];          // [1440] atomically updated, then [1440] expected
  unsigned long long* expect = acc + 1440;
  __shared__ int s_dummy[4];
  for (int i = threadIdx.x; i < 2880; i += 256) acc[i] = 0ull;
  if (threadIdx.x == 0) s_dummy[0] = 0;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    const unsigned h = hashu(blockIdx.x * 7919u + threadIdx.x * 131u + i);
    if (h & 3) atomicAdd(&acc[h % 1440u], (unsigned long long)(1 + (h >> 28)));
  }
  __syncthreads();
  // expected: thread t owns bins t, t + 256, ...: recount serially over all (thread, i)
  for (int bin = threadIdx.x; bin < 1440; bin += 256) {
    unsigned long long e = 0;
    for (int t = 0; t < 256; ++t)
      for (int i = 0; i < iters; ++i) {
        const unsigned h = hashu(blockIdx.x * 7919u + t * 131u + i);
        if ((h & 3) && (h % 1440u) == (unsigned)bin) e += (unsigned long long)(1 + (h >> 28));
      }
    expect[bin] = e;
    if (e != acc[bin]) {
      if (atomicAdd(bad, 1u) == 0) { info[0] = blockIdx.x; info[1] = bin; info[2] = (unsigned)acc[bin]; info[3] = (unsigned)e; }
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
  CK(hipMalloc(&d, 32));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  const int64_t bytes = cppf_reslayer_split_stream_bytes(256, 256, 0, 0);
  void* wq;
  CK(hipMalloc(&wq, bytes));
  CK(hipMemset(wq, 0x3c, bytes));
  for (int with_mlp = 0; with_mlp < 2; ++with_mlp)
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(d, 0, 32));
      CK(hipDeviceSynchronize());
      if (with_mlp) {
        int rc = cppf_reslayer_split(x, 256, 256, x, 256, 256, rows, wq, bytes, b, nullptr, 0, sa);
        if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
      }
      hipLaunchKernelGGL(victim_kernel, dim3(1024), dim3(256), 2880 * 8, sb, d, d + 1, 40);
      CK(hipDeviceSynchronize());
      unsigned h[5];
      CK(hipMemcpy(h, d, 20, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: %u bins wrong", with_mlp ? "beside " : "without", h[0]);
      if (h[0]) printf(" (first: block %u bin %u got %u expected %u)", h[1], h[2], h[3], h[4]);
      printf("\n");
    }
  return 0;
}
