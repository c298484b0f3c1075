// Does reslayer_split_kernel write outside its own LDS allocation?  A victim kernel (32 KiB of LDS per workgroup, fits beside
// the 125 KiB of a 256-wide MLP workgroup on the same CU) fills its LDS with a pattern and keeps re-checking it while the MLP
// kernel runs on another stream; every word that changes is counted and its LDS offset recorded.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
#include <vector>
#include <cstring>
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
// barrier victim: every wavefront publishes a value, __syncthreads(), every thread checks the other wavefronts' values
__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* first_off, unsigned* first_val, int spins) {
  extern __shared__ unsigned v_tab[];                    // padding to 32 KiB so that the placement equals the rotation kernel's
  for (int s = 0; s < spins * 40; ++s) {
    v_tab[threadIdx.x] = hashu(blockIdx.x * 7919u + s * 257u + threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const unsigned t = (threadIdx.x + 64 * k) & 255u;
      const unsigned got = v_tab[t], want = hashu(blockIdx.x * 7919u + s * 257u + t);
      if (got != want) {
        if (atomicAdd(bad, 1u) == 0) { *first_off = s; *first_val = got; }
      }
    }
    __syncthreads();
    if ((threadIdx.x >> 6) == (unsigned)(s & 3)) __builtin_amdgcn_s_sleep(8);      // skew the wavefronts
  }
}

int main(int argc, char** argv) {
  const int64_t rows = 400000;
  float *x, *b;
  CK(hipMalloc(&x, rows * 368 * 4));
  CK(hipMalloc(&b, 16 * 256 * 4));
  CK(hipMemset(x, 0, rows * 368 * 4));
  CK(hipMemset(b, 0, 16 * 256 * 4));
  unsigned* d;
  CK(hipMalloc(&d, 12));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  struct Shape { const char* name; int k, n, proj, chain; };
  const Shape shapes[] = {{"256->256 id", 256, 256, 0, 0}, {"128->256 proj chain 2", 128, 256, 1, 2}, {"256->192 proj", 256, 192, 1, 0},
                          {"360->128 proj chain 4", 360, 128, 1, 4}, {"128->64 proj", 128, 64, 1, 0}};
  for (int with_mlp = 0; with_mlp < 2; ++with_mlp)
    for (const Shape& s : shapes) {
      const int64_t bytes = cppf_reslayer_split_stream_bytes(s.k, s.n, s.proj, s.chain);
      void* wq;
      CK(hipMalloc(&wq, bytes));
      CK(hipMemset(wq, 0x3c, bytes));                 // non-zero bytes so that stray writes are visible
      CK(hipMemset(d, 0, 12));
      CK(hipDeviceSynchronize());
      if (with_mlp)
        for (int r = 0; r < 4; ++r) {
          int rc = cppf_reslayer_split(x, 368, s.k, s.proj ? x : x, 368, s.n, rows, wq, bytes, b, s.proj ? b + 2048 : nullptr, s.chain, sa);
          if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
        }
      hipLaunchKernelGGL(victim_kernel, dim3(4096), dim3(256), 32768, sb, d, d + 1, d + 2, 200);
      CK(hipDeviceSynchronize());
      unsigned h[3];
      CK(hipMemcpy(h, d, 12, hipMemcpyDeviceToHost));
      printf("%s %-24s: %u corrupted LDS words in the victim%s", with_mlp ? "beside" : "without", s.name, h[0], h[0] ? "" : "\n");
      if (h[0]) printf(" (first: byte offset %u, value 0x%08x)\n", h[1], h[2]);
      CK(hipFree(wq));
    }
  return 0;
}
