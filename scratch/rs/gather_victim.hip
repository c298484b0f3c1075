// Do 16-byte global table gathers (the rotation vote's lookup-table reads) of a small co-resident wavefront return the table's
// contents while a 448-register MLP workgroup streams its weights through the same CU's vector cache with LDS-DMA loads?
// Every thread gathers pseudo-random rows of a table whose row i is a hash of i and checks what it got.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __host__ __forceinline__ unsigned hashu(unsigned a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

__global__ void fill_kernel(int4* tab, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const unsigned h = hashu(i * 2654435761u + 11u); tab[i] = make_int4((int)h, (int)(h + 1u), (int)(h ^ 0x55aa55aau), i); }
}

__global__ __launch_bounds__(256) void victim_kernel(const int4* __restrict__ tab, int n, unsigned* bad, unsigned* info, int iters, int lds_bytes) {
  extern __shared__ char smem[];
  if (lds_bytes < 0) smem[threadIdx.x] = 0;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    const unsigned h = hashu(blockIdx.x * 7919u + threadIdx.x * 131u + i * 977u);
    const int r = (int)(h % (unsigned)n);
    const int4 e = tab[r];
    // some arithmetic between the gathers, like the vote's candidate normalisation
    const float a = (float)(h & 1023) / 512.f - 1.0f, b = (float)((h >> 10) & 1023) / 512.f - 1.0f;
    const float nn = fmaxf(__builtin_sqrtf(fmaf(b, b, a * a)), 1e-7f);
    acc += a / nn + b / nn;
    const unsigned g = hashu(r * 2654435761u + 11u);
    if ((unsigned)e.x != g || (unsigned)e.y != g + 1u || (unsigned)e.z != (g ^ 0x55aa55aau) || e.w != r) {
      if (atomicAdd(bad, 1u) == 0) { info[0] = blockIdx.x; info[1] = threadIdx.x; info[2] = i; info[3] = r; info[4] = e.x; info[5] = g; info[6] = e.w; }
    }
  }
  if (acc == 12345.678f) bad[8] = 1;
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
  const int n = 128 * 256;
  int4* tab;
  CK(hipMalloc(&tab, (size_t)n * 16));
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, tab, n);
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  const int64_t bytes = cppf_reslayer_split_stream_bytes(256, 256, 0, 0);
  void* wq;
  CK(hipMalloc(&wq, bytes));
  CK(hipMemset(wq, 0x3c, bytes));
  for (int with_mlp = 0; with_mlp < 2; ++with_mlp)
    for (int lds = 0; lds < 2; ++lds)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(d, 0, 64));
      CK(hipDeviceSynchronize());
      if (with_mlp)
        for (int r = 0; r < 2; ++r) {
          int rc = cppf_reslayer_split(x, 256, 256, x, 256, 256, rows, wq, bytes, b, nullptr, 0, sa);
          if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
        }
      hipLaunchKernelGGL(victim_kernel, dim3(4096), dim3(256), lds ? 39936 : 20000, sb, tab, n, d, d + 1, 2000, 0);
      CK(hipDeviceSynchronize());
      unsigned h[9];
      CK(hipMemcpy(h, d, 36, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel, victim LDS %d B: %u of %lld gathers returned something else", with_mlp ? "beside " : "without", lds ? 39936 : 20000, h[0], 4096ll * 256 * 2000);
      if (h[0]) printf(" (first: block %u thread %u iter %u row %u got %08x want %08x, got row tag %u)", h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
      printf("\n");
    }
  return 0;
}
