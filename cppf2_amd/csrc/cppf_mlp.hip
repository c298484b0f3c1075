// cppf_mlp.hip -- the 128-wide residual layer of the tuple / point encoders as ONE kernel on the fp32 matrix cores.
//
// The reference's tuple MLP (train_shot.py:19-73, train_dino.py:21-89) is a stack of ResLayers; five of the six layers of
// `tuple_encoder` and of `shot_encoder` are 128 -> 128 with an identity skip:
//        x <- x + relu(x W1^T + b1) W2^T (+ b2)
// As two library GEMMs that is 2 x (read x, write x) of a [rows, 128] float32 activation -- 1.3 M rows at bench size --
// per layer, with K = 128 too short for a GEMM tile to amortise its prologue (measured: 57 TFLOP/s, 0.74 ms per GEMM,
// 27 % of the whole step).  Here a wavefront takes 32 rows and keeps everything in registers:
//
//   * both products are computed TRANSPOSED, h^T = W1 x^T and y^T = x^T + W2 h^T, with v_mfma_f32_32x32x2_f32 (f32 in,
//     f32 accumulate: an exact fmaf chain, cdna_hip_programming.md), so the 32 rows of x are the COLUMNS of the
//     accumulator tiles: lane l owns row (l & 31);
//   * the K order of a sum is free, so lane half g = l >> 5 contracts over exactly the features it holds in the
//     accumulator layout (row = (reg & 3) + 8 (reg >> 2) + 4 g of a 32 x 32 tile): the 64 registers a lane loads of its
//     x row are at once the B operands of the first product and the residual accumulators of the second, and the
//     accumulators of the first product (after bias + ReLU) are the B operands of the second -- no shuffle, no LDS
//     round trip, no second pass over memory;
//   * the A operands (W1, W2: 2 x 64 KiB) live in LDS for the lifetime of the persistent workgroup, stored so that
//     one ds_read_b128 yields a lane's values for the four 32-row output tiles of a K step.
// Per 32 rows: 16 + 16 global 16-byte accesses per lane, 128 LDS reads, 512 MFMAs (= 32 768 cycles of the SIMD's matrix
// pipe); two wavefronts per SIMD cover each other's loads.  MFMA-bound: 84 GFLOP per layer.
#include "cppf_common.h"
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RL_DIM 128
#define RL_THREADS 512                 // 8 wavefronts, 2 per SIMD
#define RL_ROWS_PER_WAVE 32

// feature handled by lane half g at K step s = 16 t' + 4 q' + c'  (also: accumulator register (t', 4 q' + c') of that lane)
__device__ __host__ __forceinline__ int rl_feature(int g, int s) {
  return 32 * (s >> 4) + 8 * ((s >> 2) & 3) + 4 * g + (s & 3);
}

template <int S>
__device__ __forceinline__ float rl_reg(const f32x16 (&v)[4]) {
  return v[S >> 4][S & 15];
}

// one transposed product: acc[t] += W[32 t + (l & 31)][feature(g, s)] * b(s) over the 64 K steps.  The A values of
// step s + 2 are requested from LDS before the four MFMAs of step s issue (their 256 cycles cover the read), and a
// scheduling barrier per step keeps the compiler from hoisting all 64 reads to the front (256 live registers, spills).
template <int S>
struct RlSteps {
  static __device__ __forceinline__ void run(const f32x4* __restrict__ w_lds, const f32x16 (&b)[4], f32x16 (&acc)[4],
                                             f32x4 a0, f32x4 a1) {
    f32x4 a2 = a1;
    if (S + 2 < 64) a2 = w_lds[(S + 2) * 32];
    const float bv = rl_reg<S>(b);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bv, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bv, acc[3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    RlSteps<S + 1>::run(w_lds, b, acc, a1, a2);
  }
};
template <>
struct RlSteps<64> {
  static __device__ __forceinline__ void run(const f32x4* __restrict__, const f32x16 (&)[4], f32x16 (&)[4], f32x4, f32x4) {}
};

__device__ __forceinline__ void rl_product(const f32x4* __restrict__ w_lds, const f32x16 (&b)[4], f32x16 (&acc)[4]) {
  const f32x4 a0 = w_lds[0], a1 = w_lds[32];
  RlSteps<0>::run(w_lds, b, acc, a0, a1);
}

__global__ __launch_bounds__(RL_THREADS, 1) void reslayer128_kernel(float* __restrict__ x, int64_t rows,
                                                                    const float* __restrict__ w1,
                                                                    const float* __restrict__ b1,
                                                                    const float* __restrict__ w2) {
  // [matrix][g][s][i & 31] -> float4 over the four output tiles t: W[32 t + (i & 31)][feature(g, s)]
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  f32x4* s_w4 = reinterpret_cast<f32x4*>(s_w);
  float* s_b = s_w + 2 * RL_DIM * RL_DIM;              // bias, 128 floats
  // coalesced reads of the two row-major matrices, scattered into the K-step-major LDS layout: element W[i][f] belongs to
  // lane half g = (f >> 2) & 1 at step s = 16 (f >> 5) + 4 ((f >> 3) & 3) + (f & 3), slot (i & 31), tile i >> 5
  for (int e = threadIdx.x; e < 2 * RL_DIM * RL_DIM; e += RL_THREADS) {
    const int m = e >> 14, i = (e >> 7) & 127, f = e & 127;
    const int g = (f >> 2) & 1, s = 16 * (f >> 5) + 4 * ((f >> 3) & 3) + (f & 3);
    s_w[((((m * 2 + g) * 64 + s) * 32 + (i & 31)) << 2) + (i >> 5)] = (m ? w2 : w1)[(i << 7) + f];
  }
  if (threadIdx.x < RL_DIM) s_b[threadIdx.x] = b1[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, g = lane >> 5;
  const f32x4* w1_lds = s_w4 + (0 * 2 + g) * 64 * 32 + r;
  const f32x4* w2_lds = s_w4 + (1 * 2 + g) * 64 * 32 + r;
  const int64_t tiles = (rows + RL_ROWS_PER_WAVE - 1) / RL_ROWS_PER_WAVE;
  const int64_t stride = (int64_t)gridDim.x * (RL_THREADS / 64);
  // the next tile's rows are requested while the current tile is in the matrix pipe (register double buffer)
  f32x4 nx[16];
  auto fetch = [&](int64_t tile) {
    const int64_t row = tile * RL_ROWS_PER_WAVE + r;
    const bool in = tile < tiles && row < rows;
    const float* xp = x + (in ? row : 0) * RL_DIM + 4 * g;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (in) v = *reinterpret_cast<const f32x4*>(xp + 32 * (k >> 2) + 8 * (k & 3));
      nx[k] = v;
    }
  };
  const int64_t first = (int64_t)blockIdx.x * (RL_THREADS / 64) + wave;
  fetch(first);
  for (int64_t tile = first; tile < tiles; tile += stride) {
    const int64_t row = tile * RL_ROWS_PER_WAVE + r;
    const bool in = row < rows;
    float* xp = x + (in ? row : 0) * RL_DIM + 4 * g;
    // this lane's 64 features of its row: register (t, 4 q + c) <- x[row][32 t + 8 q + 4 g + c]
    f32x16 xr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = nx[4 * t + q];
        xr[t][4 * q + 0] = v.x; xr[t][4 * q + 1] = v.y; xr[t][4 * q + 2] = v.z; xr[t][4 * q + 3] = v.w;
      }
    }
    fetch(tile + stride);
    // h^T = relu(W1 x^T + b1): accumulators start at the bias of their feature
    f32x16 h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(s_b + 32 * t + 8 * q + 4 * g);
        h[t][4 * q + 0] = bv.x; h[t][4 * q + 1] = bv.y; h[t][4 * q + 2] = bv.z; h[t][4 * q + 3] = bv.w;
      }
    }
    rl_product(w1_lds, xr, h);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) h[t][e] = (h[t][e] < 0.0f) ? 0.0f : h[t][e];     // relu; NaN stays NaN like torch.relu
    }
    // y^T = x^T + W2 h^T: the x registers are the accumulators
    rl_product(w2_lds, h, xr);
    if (in) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v.x = xr[t][4 * q + 0]; v.y = xr[t][4 * q + 1]; v.z = xr[t][4 * q + 2]; v.w = xr[t][4 * q + 3];
          *reinterpret_cast<f32x4*>(xp + 32 * t + 8 * q) = v;
        }
      }
    }
  }
}

// x float32 [rows, 128] (device, contiguous), updated in place: x <- x + relu(x W1^T + b1) W2^T.  w1 / w2 float32 [128, 128]
// row-major [out, in] (the nn.Linear weights), b1 float32 [128].  The second layer's bias is the caller's (it commutes
// with the residual stream: cppf2_amd.models.fused_stack carries it as a pending offset).
extern "C" int cppf_reslayer128(float* x, int64_t rows, const float* w1, const float* b1, const float* w2, void* stream) {
  CPPF_CHECK_ARG(x && w1 && b1 && w2 && rows >= 0);
  if (rows == 0) return CPPF_OK;
  const int lds_bytes = (2 * RL_DIM * RL_DIM + RL_DIM) * 4;
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  CPPF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev & 63] == 0) {
      hipDeviceProp_t prop;
      CPPF_HIP(hipGetDeviceProperties(&prop, dev));
      CPPF_HIP(hipFuncSetAttribute((const void*)reslayer128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      cus[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
  }
  const int64_t tiles = (rows + RL_ROWS_PER_WAVE - 1) / RL_ROWS_PER_WAVE;
  int64_t blocks = (tiles + (RL_THREADS / 64) - 1) / (RL_THREADS / 64);
  if (blocks > cus[dev & 63]) blocks = cus[dev & 63];
  hipLaunchKernelGGL(reslayer128_kernel, dim3((unsigned)blocks), dim3(RL_THREADS), lds_bytes, (hipStream_t)stream, x, rows,
                     w1, b1, w2);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The narrow output ResLayer of the scale head (train_shot.py:67-71: ResLayer(64, 3)) -- too narrow for a matrix-core tile --
// as plain float32 arithmetic, one thread per row, with the scatter of the kept pairs' rows folded into the store:
//     y[n] = (b0[n] + sum_k x[k] W0[n][k]) + sum_m relu(b1[m] + sum_k x[k] W1[m][k]) W2[n][m]
// every sum an fmaf chain in index order (a float32 GEMM up to the summation order; results do not depend on the row count or
// on the row's position, unlike a library GEMM's tiling).  The weights are read through wave-uniform addresses (scalar loads).
// ---------------------------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void reslayer_tail_kernel(const float* __restrict__ x, int64_t ldx, int k_in, int64_t rows,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w2, const int32_t* __restrict__ scatter_rows,
                                                            const int32_t* __restrict__ valid_count, int per_group,
                                                            float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  int64_t dst = i;
  if (scatter_rows) {
    if (valid_count && (int)(i % per_group) >= valid_count[i / per_group]) return;     // a padded entry of its group
    dst = scatter_rows[i];
  }
  const float* xr = x + i * ldx;
  float h[N], s[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    h[n] = b1[n];
    s[n] = w0 ? b0[n] : xr[n];
  }
  for (int k = 0; k < k_in; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float* a = w1 + (int64_t)n * k_in + k;
      h[n] = fmaf(v.w, a[3], fmaf(v.z, a[2], fmaf(v.y, a[1], fmaf(v.x, a[0], h[n]))));
      if (w0) {
        const float* c = w0 + (int64_t)n * k_in + k;
        s[n] = fmaf(v.w, c[3], fmaf(v.z, c[2], fmaf(v.y, c[1], fmaf(v.x, c[0], s[n]))));
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) h[n] = (h[n] < 0.0f) ? 0.0f : h[n];        // NaN stays NaN like torch.relu
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float y = s[n];
#pragma unroll
    for (int m = 0; m < N; ++m) y = fmaf(h[m], w2[n * N + m], y);
    out[dst * ldo + n] = y;
  }
}

// out[r] = skip(x[i]) + relu(x[i] W1^T + b1) W2^T for a ResLayer with n_out <= 8 outputs: x float32 [rows, >= k_in] (row stride
// ldx, a multiple of 4; 16-byte aligned; k_in a multiple of 4), w1 / w0 float32 [n_out, k_in] and w2 float32 [n_out, n_out] as
// nn.Linear stores them, b1 / b0 float32 [n_out] (b0 carries fc2's bias too, as everywhere in this library; w0 == b0 == NULL:
// identity skip, k_in == n_out).  r = i, or scatter_rows[i] when given; with valid_count (int32 [rows / per_group]) entry i
// is skipped unless (i % per_group) < valid_count[i / per_group] -- the padded tail of each scene's kept-pair list
// (cppf_kept_rows32): only real pairs are written, each exactly once.
extern "C" int cppf_reslayer_tail(const float* x, int64_t ldx, int32_t k_in, int32_t n_out, int64_t rows, const float* w1,
                                  const float* b1, const float* w0, const float* b0, const float* w2,
                                  const int32_t* scatter_rows, const int32_t* valid_count, int32_t per_group, float* out,
                                  int64_t ldo, void* stream) {
  CPPF_CHECK_ARG(x && w1 && b1 && w2 && out && rows >= 0 && n_out >= 1 && n_out <= 8 && ldo >= n_out);
  CPPF_CHECK_ARG(k_in > 0 && (k_in & 3) == 0 && ldx >= k_in && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0);
  CPPF_CHECK_ARG((w0 == nullptr) == (b0 == nullptr) && (w0 != nullptr || k_in == n_out));
  CPPF_CHECK_ARG(valid_count == nullptr || (scatter_rows != nullptr && per_group > 0 && rows % per_group == 0));
  if (rows == 0) return CPPF_OK;
  const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define CPPF_TAIL(N)                                                                                                    \
  case N:                                                                                                               \
    hipLaunchKernelGGL(reslayer_tail_kernel<N>, grid, block, 0, st, x, ldx, k_in, rows, w1, b1, w0, b0, w2, scatter_rows, \
                       valid_count, per_group, out, ldo);                                                               \
    break;
  switch (n_out) {
    CPPF_TAIL(1) CPPF_TAIL(2) CPPF_TAIL(3) CPPF_TAIL(4) CPPF_TAIL(5) CPPF_TAIL(6) CPPF_TAIL(7) CPPF_TAIL(8)
  }
#undef CPPF_TAIL
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
