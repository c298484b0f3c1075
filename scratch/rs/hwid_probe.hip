// Where do the workgroups of a 2-per-CU launch land?  256 threads, 74 KiB of LDS, grid 512: HW_ID / XCC_ID / start time per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ char lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)(t0 & 0xffffffffu);
    out[4 * blockIdx.x + 3] = (unsigned)(t0 >> 32);
  }
  lds[threadIdx.x] = (char)spin;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
  if (lds[threadIdx.x ^ 1] == 77) out[0] = 0;
}
int main() {
  unsigned* d;
  const int G = 512;
  hipMalloc(&d, G * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024);
  hipLaunchKernelGGL(probe, dim3(G), dim3(256), 74 * 1024, 0, d, 50);
  hipDeviceSynchronize();
  std::vector<unsigned> h(G * 4);
  hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int i = 0; i < G; ++i) { unsigned long long t = ((unsigned long long)h[4 * i + 3] << 32) | h[4 * i + 2]; if (t < tmin) tmin = t; }
  for (int i = 0; i < G; i += 1) {
    const unsigned hw = h[4 * i];
    unsigned long long t = ((unsigned long long)h[4 * i + 3] << 32) | h[4 * i + 2];
    if (i < 40 || (i >= 250 && i < 275))
      printf("wg %3d  wave_id %2u simd %u pipe %u cu %2u sh %u se %u  xcc %u  t %llu\n", i, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15,
             (hw >> 12) & 1, (hw >> 13) & 7, h[4 * i + 1] & 15, t - tmin);
  }
  // how many workgroups share a (xcc, se, sh, cu), and their wave ids
  int odd = 0;
  for (int i = 0; i < G; ++i) odd += h[4 * i] & 1;
  printf("workgroups with an odd wave slot: %d of %d\n", odd, G);
  return 0;
}
