// Packed float32 instructions next to ordinary VALU instructions that touch the same registers -- the three dependences the
// compiler's SLP-vectorised code contains (rot_bins_lut_kernel: v_pk_add_f32 v[14:15] ... v_mov_b32 v15; v_pk_add_f32 reading
// v[6:7] ... v_cndmask_b32 v6) -- while a 448-register MFMA wavefront of another stream shares the SIMD:
//   RAW: v_pk_mul_f32 writes v[40:41]; one wait state; v_add_f32 reads v40, v41
//   WAW: v_pk_add_f32 writes v[40:41]; the next VALU instruction overwrites v41; v41 is read later
//   WAR: v_pk_add_f32 reads v[42:43]; the next VALU instruction overwrites v42; the packed result is read later
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
thread_local char g_cppf_err[256];
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void victim_kernel(unsigned* bad, int iters) {
  extern __shared__ char smem[];
  float a = 1.0f + (float)threadIdx.x * 0.00390625f + (float)(blockIdx.x & 1023) * 1e-4f, b = 0.5f + (float)threadIdx.x * 0.001f;
  unsigned nraw = 0, nwaw = 0, nwar = 0;
  for (int i = 0; i < iters; ++i) {
    const float c = a * 0.75f, d = b + 2.0f;
    float raw, waw, war0, war1;
    asm volatile("v_mov_b32 v42, %1\n v_mov_b32 v43, %2\n v_mov_b32 v44, %3\n v_mov_b32 v45, %4\n s_nop 4\n"
                 "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n s_nop 0\n v_add_f32 %0, v40, v41"
                 : "=&v"(raw) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41", "v42", "v43", "v44", "v45");
    nraw += (__float_as_uint(raw) != __float_as_uint(a * c + b * d)) ? 1u : 0u;
    asm volatile("v_mov_b32 v42, %1\n v_mov_b32 v43, %2\n v_mov_b32 v44, %3\n v_mov_b32 v45, %4\n s_nop 4\n"
                 "v_pk_add_f32 v[40:41], v[42:43], v[44:45]\n v_mov_b32 v41, %1\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v41"
                 : "=&v"(waw) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41", "v42", "v43", "v44", "v45");
    nwaw += (__float_as_uint(waw) != __float_as_uint(a)) ? 1u : 0u;
    asm volatile("v_mov_b32 v42, %2\n v_mov_b32 v43, %3\n v_mov_b32 v44, %4\n v_mov_b32 v45, %5\n s_nop 4\n"
                 "v_pk_add_f32 v[40:41], v[42:43], v[44:45]\n v_mov_b32 v42, 1.0\n v_mov_b32 v45, 2.0\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v40\n v_mov_b32 %1, v41"
                 : "=&v"(war0), "=&v"(war1) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41", "v42", "v43", "v44", "v45");
    nwar += (__float_as_uint(war0) != __float_as_uint(a + c) || __float_as_uint(war1) != __float_as_uint(b + d)) ? 1u : 0u;
    a = a * 1.0009765625f + 0.0625f;
    a = (a > 1000.0f) ? a * 0.0009765625f : a;
    b = b * 0.99951171875f + 0.03125f;
  }
  if (nraw | nwaw | nwar) { atomicAdd(&bad[0], nraw); atomicAdd(&bad[1], nwaw); atomicAdd(&bad[2], nwar); }
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
      hipLaunchKernelGGL(victim_kernel, dim3(8192), dim3(256), 34000, sb, d, 4000);
      CK(hipDeviceSynchronize());
      unsigned h[4];
      CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
      printf("%s the 256-wide MLP kernel: of %lld evaluations each, wrong: read-after-write %u, write-after-write %u, write-after-read %u\n",
             with_mlp ? "beside " : "without", 8192ll * 256 * 4000, h[0], h[1], h[2]);
    }
  return 0;
}
