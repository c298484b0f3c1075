// Test-only kernels for the wavefront idioms of cppf2_amd/csrc/cppf_common.h (built and run by tests/test_wave_idioms_gpu.py):
// each idiom against the portable form it replaces, and sqrt_rn against the compiler's correctly rounded sqrtf over EVERY float32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cppf_common.h"

// out[0]: mismatches of sqrt_rn vs sqrtf over all 2^32 bit patterns (NaN results compare as equal); out[1]: a witness pattern
__global__ void sqrt_all_kernel(unsigned long long* out) {
  unsigned long long bad = 0;
  uint32_t witness = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)i);
    const float a = sqrt_rn(x), b = __builtin_sqrtf(x);
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    if (!same) { ++bad; witness = (uint32_t)i; }
  }
  if (bad) { atomicAdd(&out[0], bad); out[1] = witness; }
}

// one wavefront per block over rows of 64 values: every idiom's result next to the portable form's
__global__ void idioms_kernel(const uint32_t* in, const double* din, int rows, unsigned long long* bad) {
  const int lane = threadIdx.x;
  unsigned long long b[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per idiom
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const uint32_t v = in[(size_t)r * 64 + lane];
    const double d = din[(size_t)r * 64 + lane];
    // ballot / lanes_below
    const bool p = (v & 1u) != 0u;
    const unsigned long long m0 = __ballot(p), m1 = wave_ballot(p);
    b[0] += (m0 != m1);
    b[1] += (lanes_below(m1) != __popcll(m0 & ((1ull << lane) - 1ull)));
    // inclusive scan
    uint32_t ref = v & 0xffffu;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(ref, off); if (lane >= off) ref += o; }
    b[2] += (wave_inclusive_scan_u32(v & 0xffffu) != ref);
    // upper_half (result in lanes < 32; called with the whole wavefront active, as the exchange needs), 32- and 64-bit
    {
      const uint32_t uv = upper_half(v);
      const double ud = upper_half(d);
      const uint32_t rv = (uint32_t)__shfl((int)v, (lane + 32) & 63);
      const double rd = __shfl(d, (lane + 32) & 63);
      if (lane < 32) {
        b[3] += (uv != rv);
        b[4] += (__double_as_longlong(ud) != __double_as_longlong(rd));
      }
    }
    // row_down by 1 and 2 inside rows of 16 (lanes without a source: 0.0)
    {
      const double s1 = __shfl(d, min(lane + 1, 63)), s2 = __shfl(d, min(lane + 2, 63));
      const double w1 = ((lane & 15) + 1 < 16) ? s1 : 0.0, w2 = ((lane & 15) + 2 < 16) ? s2 : 0.0;
      b[5] += (__double_as_longlong(row_down<0x101>(d)) != __double_as_longlong(w1));
      b[6] += (__double_as_longlong(row_down<0x102>(d)) != __double_as_longlong(w2));
    }
    // wave_sum: bit-identical on every lane to the xor butterfly in the order 1, 2, 4, 8, 16, 32
    {
      double t = d;
      for (int off = 1; off < 64; off <<= 1) t += __shfl_xor(t, off);
      b[7] += (__double_as_longlong(wave_sum(d)) != __double_as_longlong(t));
    }
  }
  for (int k = 0; k < 8; ++k)
    if (b[k]) atomicAdd(&bad[k], b[k]);
}

extern "C" int check_sqrt_all(unsigned long long* host_out) {
  unsigned long long* d;
  if (hipMalloc(&d, 16) != hipSuccess) return -1;
  hipMemset(d, 0, 16);
  hipLaunchKernelGGL(sqrt_all_kernel, dim3(4096), dim3(256), 0, 0, d);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  hipMemcpy(host_out, d, 16, hipMemcpyDeviceToHost);
  hipFree(d);
  return 0;
}

extern "C" int check_idioms(const uint32_t* host_in, const double* host_din, int rows, unsigned long long* host_bad) {
  uint32_t* in; double* din; unsigned long long* bad;
  if (hipMalloc(&in, (size_t)rows * 256) != hipSuccess || hipMalloc(&din, (size_t)rows * 512) != hipSuccess || hipMalloc(&bad, 64) != hipSuccess) return -1;
  hipMemcpy(in, host_in, (size_t)rows * 256, hipMemcpyHostToDevice);
  hipMemcpy(din, host_din, (size_t)rows * 512, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 64);
  hipLaunchKernelGGL(idioms_kernel, dim3(256), dim3(64), 0, 0, in, din, rows, bad);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  hipMemcpy(host_bad, bad, 64, hipMemcpyDeviceToHost);
  hipFree(in); hipFree(din); hipFree(bad);
  return 0;
}
