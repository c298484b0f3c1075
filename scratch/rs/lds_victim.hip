// Does reslayer_split_kernel write outside its own LDS allocation?  A victim kernel (32 KiB of LDS per workgroup, fits beside
// the 125 KiB of a 256-wide MLP workgroup on the same CU) fills its LDS with a pattern and keeps re-checking it while the MLP
// kernel runs on another stream; every word that changes is counted and its LDS offset recorded.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
#include <vector>
#include <cstring>
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
// mixed victim (the rotation-vote kernel's phase 2): random-index 16-byte and 8-byte LDS reads, each checked, interleaved with
// 64-bit LDS atomics to another region from every wavefront; the atomic totals are checked at the end
__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* first_off, unsigned* first_val, int spins) {
  extern __shared__ uint4 v_tab[];                       // 1280 entries = 20 KiB, then 1440 x 8 B of accumulators (11.25 KiB)
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(v_tab + 1280);
  for (int i = threadIdx.x; i < 1280; i += 256) v_tab[i] = make_uint4(0xA5000000u ^ i, 0x5A000000u ^ i, 0x3C000000u ^ i, 0xC3000000u ^ i);
  for (int i = threadIdx.x; i < 1440; i += 256) acc[i] = 0ull;
  __syncthreads();
  const unsigned long long* v64 = reinterpret_cast<const unsigned long long*>(v_tab);
  unsigned long long mine = 0;
  for (int s = 0; s < spins * 16; ++s) {
    const unsigned h = hashu(blockIdx.x * 7919u + threadIdx.x * 131u + s);
    const unsigned i = h % 1280u, j = (h >> 11) % 2560u;
    const uint4 q = v_tab[i];
    const unsigned long long w = v64[j];
    const unsigned long long want = (j & 1) ? (((unsigned long long)(0xC3000000u ^ (j >> 1)) << 32) | (0x3C000000u ^ (j >> 1)))
                                            : (((unsigned long long)(0x5A000000u ^ (j >> 1)) << 32) | (0xA5000000u ^ (j >> 1)));
    if (q.x != (0xA5000000u ^ i) || q.y != (0x5A000000u ^ i) || q.z != (0x3C000000u ^ i) || q.w != (0xC3000000u ^ i) || w != want) {
      if (atomicAdd(bad, 1u) == 0) { *first_off = i * 16; *first_val = q.x; }
    }
    if (h & 0x30000000u) {
      const unsigned long long v = 1 + (h >> 30);
      atomicAdd(&acc[(h >> 4) % 1440u], v);
      mine += v;
    }
  }
  // total of the accumulators == total of what the threads added
  __shared__ unsigned long long s_tot[2];
  if (threadIdx.x == 0) { s_tot[0] = 0; s_tot[1] = 0; }
  __syncthreads();
  atomicAdd(&s_tot[0], mine);
  unsigned long long part = 0;
  for (int i = threadIdx.x; i < 1440; i += 256) part += acc[i];
  atomicAdd(&s_tot[1], part);
  __syncthreads();
  if (threadIdx.x == 0 && s_tot[0] != s_tot[1]) {
    if (atomicAdd(bad, 1u) == 0) { *first_off = 0xffffffffu; *first_val = (unsigned)(s_tot[1] - s_tot[0]); }
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
