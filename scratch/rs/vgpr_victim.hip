// Do the VGPRs of a small co-resident wavefront stay intact while a 448-register MLP wavefront runs on the same SIMD?
// Each victim thread keeps NV values live in registers across a spin loop and checks them at the end.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define NV 26
__device__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, unsigned* info, int spins) {
  unsigned v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = hashu(blockIdx.x * 977u + threadIdx.x * 131u + i);
  for (int s = 0; s < spins; ++s) {
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));        // every value stays in a VGPR across the loop
    __builtin_amdgcn_s_sleep(64);
    // a little VALU work on the live values (self-inverse: xor twice)
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] ^= (unsigned)s;
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] ^= (unsigned)s;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned want = hashu(blockIdx.x * 977u + threadIdx.x * 131u + i);
    if (v[i] != want) {
      if (atomicAdd(bad, 1u) == 0) { info[0] = blockIdx.x; info[1] = threadIdx.x; info[2] = i; info[3] = v[i]; info[4] = want; }
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
      hipLaunchKernelGGL(victim_kernel, dim3(4096), dim3(256), 0, sb, d, d + 1, 300);
      CK(hipDeviceSynchronize());
      unsigned h[6];
      CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: %u register values wrong", with_mlp ? "beside " : "without", h[0]);
      if (h[0]) printf(" (first: block %u thread %u reg %u got %08x expected %08x)", h[1], h[2], h[3], h[4], h[5]);
      printf("\n");
    }
  return 0;
}
