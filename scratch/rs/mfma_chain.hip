// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 in dependent chains of length L on one wave per SIMD,
// optionally with F independent VALU fillers between consecutive MFMAs.  Prints shader cycles per MFMA (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int L, int F, int NACC>
__global__ __launch_bounds__(256, 1) void chain_kernel(const u32x4* in, float* out, int iters, long long* cyc) {
  u32x4 a = in[threadIdx.x], b = in[256 + threadIdx.x];
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float f0 = out[threadIdx.x], f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int k = 0; k < L; ++k) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < F; ++f) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f2) : "v"(f3));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = f0 + f2;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int L, int F, int NACC>
void run(const char* name, const u32x4* in, float* out, long long* cyc, int grid) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  chain_kernel<L, F, NACC><<<grid, 256>>>(in, out, 10, cyc);
  hipEventRecord(e0);
  chain_kernel<L, F, NACC><<<grid, 256>>>(in, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * L * NACC;
  printf("%-34s grid %4d: %.1f memtime ticks / MFMA, %.2f ns / MFMA  (%.0f TF/s chip-equivalent at this grid)\n", name, grid, c / n, ms * 1e6 / n,
         grid * 4.0 * 32768.0 * n / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const bool zero = argc > 1 && argv[1][0] == 'z';
  u32x4* in; float* out; long long* cyc;
  hipMalloc(&in, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  unsigned h[2048];
  unsigned s = 12345;
  for (int i = 0; i < 2048; ++i) { s = s * 1664525u + 1013904223u; unsigned e = 0x3f80 + ((s >> 8) & 0x7f); unsigned e2 = 0x3f80 + ((s >> 20) & 0x7f); h[i] = zero ? 0 : (e | (e2 << 16)) ^ ((s & 1) << 15); }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(out, 0, 256 * 256 * 4);
  for (int grid : {1, 256}) {
    run<6, 0, 8>("chain 6, no filler, 8 acc", in, out, cyc, grid);
    run<1, 0, 8>("independent (chain 1), 8 acc", in, out, cyc, grid);
    run<2, 0, 8>("chain 2, 8 acc", in, out, cyc, grid);
    run<6, 1, 8>("chain 6, 2 VALU between each", in, out, cyc, grid);
    run<6, 2, 8>("chain 6, 4 VALU between each", in, out, cyc, grid);
    run<6, 3, 8>("chain 6, 6 VALU between each", in, out, cyc, grid);
    run<1, 1, 8>("independent, 2 VALU between", in, out, cyc, grid);
    run<1, 2, 8>("independent, 4 VALU between", in, out, cyc, grid);
    run<1, 3, 8>("independent, 6 VALU between", in, out, cyc, grid);
    run<1, 4, 8>("independent, 8 VALU between", in, out, cyc, grid);
  }
  return 0;
}
