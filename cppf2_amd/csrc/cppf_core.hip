// cppf_core.hip -- sampler, scene bounds, tuple encode, bin decode, vote-parameter decode.
// gfx950 only.  See include/cppf_hip.h for the contract of each entry point.
#include "cppf_common.h"
#include <hip/hip_fp16.h>

thread_local char g_cppf_err[256] = "";

extern "C" int cppf_version(void) { return CPPF_ABI_VERSION; }
extern "C" const char* cppf_last_error_string(void) { return g_cppf_err; }

// scene lookup for flattened rows: largest b with off[b] <= row (B is small: linear/binary search in L1)
__device__ __forceinline__ int find_scene(const int32_t* __restrict__ off, int B, int64_t row) {
  int lo = 0, hi = B;   // invariant: off[lo] <= row < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)off[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------
// a1. tuple sampler
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_tuples_kernel(int B, const int32_t* __restrict__ pt_off,
                                                            const int32_t* __restrict__ tup_off, int k,
                                                            uint32_t key0, uint32_t key1, int32_t sid_base,
                                                            int32_t sid_stride, int32_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const uint32_t n = (uint32_t)(pt_off[b + 1] - pt_off[b]);
  const uint32_t sid = (uint32_t)(sid_base + b * sid_stride);
  // a workgroup's 256 rows are contiguous in the table: staged in LDS and written out as consecutive words (a row is 8 .. 32
  // bytes: row-wise stores touch every cache line of the tile k times)
  __shared__ int32_t s_rows[256 * 8];
  for (int tb = blockIdx.x * blockDim.x; tb < nt; tb += gridDim.x * blockDim.x) {
    const int t = tb + threadIdx.x;
    if (t < nt) {
      for (int blk = 0; blk * 4 < k; ++blk) {
        const Philox4 w = philox4x32_10((uint32_t)t, (uint32_t)blk, sid, 0u, key0, key1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (blk * 4 + j < k) s_rows[threadIdx.x * k + blk * 4 + j] = (int32_t)(((uint64_t)w.v[j] * n) >> 32);
      }
    }
    __syncthreads();
    const int words = min(256, nt - tb) * k;
    int32_t* dst = out + (int64_t)(t0 + tb) * k;
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = s_rows[i];
    __syncthreads();
  }
}

extern "C" int cppf_sample_tuples(int B, const int32_t* pt_off, const int32_t* tup_off, int max_t, int k,
                                  uint64_t seed, int32_t scene_id_base, int32_t scene_id_stride,
                                  int32_t* out_idx, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pt_off && tup_off && out_idx);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  if (max_t <= 0) return CPPF_OK;
  dim3 grid((max_t + 255) / 256, B);
  hipLaunchKernelGGL(sample_tuples_kernel, grid, dim3(256), 0, (hipStream_t)stream, B, pt_off, tup_off, k,
                     (uint32_t)seed, (uint32_t)(seed >> 32), scene_id_base, scene_id_stride, out_idx);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

__global__ __launch_bounds__(256) void philox_uniform_kernel(int B, const int32_t* __restrict__ tup_off, int m,
                                                             uint32_t key0, uint32_t key1, int32_t sid_base,
                                                             int32_t sid_stride, uint32_t stream_id,
                                                             float* __restrict__ out) {
  const int b = blockIdx.y;
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const uint32_t sid = (uint32_t)(sid_base + b * sid_stride);
  __shared__ float s_rows[256 * 8];               // (staged like sample_tuples_kernel's rows: consecutive words out)
  for (int tb = blockIdx.x * blockDim.x; tb < nt; tb += gridDim.x * blockDim.x) {
    const int t = tb + threadIdx.x;
    if (t < nt) {
      for (int blk = 0; blk * 4 < m; ++blk) {
        const Philox4 w = philox4x32_10((uint32_t)t, (uint32_t)blk, sid, stream_id, key0, key1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (blk * 4 + j < m) s_rows[threadIdx.x * m + blk * 4 + j] = (float)(w.v[j] >> 8) * 5.9604644775390625e-8f;  // 2^-24
      }
    }
    __syncthreads();
    const int words = min(256, nt - tb) * m;
    float* dst = out + (int64_t)(t0 + tb) * m;
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = s_rows[i];
    __syncthreads();
  }
}

extern "C" int cppf_philox_uniform(int B, const int32_t* tup_off, int max_t, int m, uint64_t seed,
                                   int32_t scene_id_base, int32_t scene_id_stride, int32_t stream_id,
                                   float* out, void* stream) {
  CPPF_CHECK_ARG(B > 0 && tup_off && out);
  CPPF_CHECK_ARG(m >= 1 && m <= 8);
  if (max_t <= 0) return CPPF_OK;
  dim3 grid((max_t + 255) / 256, B);
  hipLaunchKernelGGL(philox_uniform_kernel, grid, dim3(256), 0, (hipStream_t)stream, B, tup_off, m,
                     (uint32_t)seed, (uint32_t)(seed >> 32), scene_id_base, scene_id_stride,
                     (uint32_t)stream_id, out);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// x[np.isnan(x)] = 0 in place (eval.py:215-216, on the normals; the descriptors get it inside cppf_shot_describe).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nan_to_zero_kernel(float* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (!(v == v)) x[i] = 0.0f;
  }
}

extern "C" int cppf_nan_to_zero(float* x, int64_t n, void* stream) {
  CPPF_CHECK_ARG((x != nullptr || n == 0) && n >= 0);
  if (n == 0) return CPPF_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(nan_to_zero_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, n);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// float32 -> float16 (round to nearest even): the per-point feature table of BASELINE config 5 ("fp16 features") as
// cppf_encode_tuples_shot_f16 gathers it.
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (_Float16)x[i];
}

extern "C" int cppf_cast_f16(const float* x, void* out_half, int64_t n, void* stream) {
  CPPF_CHECK_ARG(((x != nullptr && out_half != nullptr) || n == 0) && n >= 0);
  if (n == 0) return CPPF_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(cast_f16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                     (_Float16*)out_half, n);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// scene bounds (train_dino.py:172-173): one workgroup per scene, min/max over the cloud.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scene_bounds_kernel(const float* __restrict__ pts,
                                                           const int32_t* __restrict__ pt_off, float res,
                                                           CppfSceneGrid* __restrict__ out) {
  const int b = blockIdx.x;
  const int p0 = pt_off[b], n = pt_off[b + 1] - p0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = pts[3 * (int64_t)(p0 + i) + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
  __shared__ float s_mn[4][3], s_mx[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
    }
  }
  const int w = threadIdx.x >> 6;
  if (wave_lane() == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_mn[w][c] = mn[c]; s_mx[w][c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    CppfSceneGrid g;
    int64_t cells = 1;
    g.flags = (n <= 0) ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float lo = fminf(fminf(s_mn[0][c], s_mn[1][c]), fminf(s_mn[2][c], s_mn[3][c]));
      const float hi = fmaxf(fmaxf(s_mx[0][c], s_mx[1][c]), fmaxf(s_mx[2][c], s_mx[3][c]));
      g.c0[c] = lo;
      const float q = (hi - lo) / res;                 // float32 division, then .long() truncation
      int gi = (n > 0 && q < 2.0e9f) ? (int)q + 1 : 0;
      g.g[c] = gi;
      cells *= (int64_t)gi;
    }
    if (cells > 0x7fffffffLL) { g.flags |= 2; cells = 0; }
    if (n <= 0) cells = 0;
    g.ncell = (int32_t)cells;
    out[b] = g;
  }
}

extern "C" int cppf_scene_bounds(int B, const float* pts, const int32_t* pt_off, float res, CppfSceneGrid* out,
                                 void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out);
  CPPF_CHECK_ARG(res > 0.0f);
  hipLaunchKernelGGL(scene_bounds_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pts, pt_off, res, out);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// a3. tuple encode.  One thread per float4 of an output row: rows are 16-byte aligned whenever
// row_len % 4 == 0 (360 for the SHOT model), so every store is a coalesced dwordx4; the feature
// table [n, feat_dim] is small (1 MB/scene) and served by L2.
// ---------------------------------------------------------------------------------------------
// q-th pair of itertools.combinations(range(k), 2), tabulated on the host (k <= 8 -> at most 28 pairs)
struct ComboTable {
  int8_t i[28];
  int8_t j[28];
};

static inline ComboTable make_combos(int k) {
  ComboTable c;
  int q = 0;
  for (int i = 0; i < k; ++i)
    for (int j = i + 1; j < k; ++j) { c.i[q] = (int8_t)i; c.j[q] = (int8_t)j; ++q; }
  for (; q < 28; ++q) { c.i[q] = 0; c.j[q] = 0; }
  return c;
}

__device__ __forceinline__ float encode_scalar(const float* __restrict__ pts, const float* __restrict__ nrm,
                                               const int32_t* __restrict__ row_idx, int p0, int np, int f,
                                               const ComboTable& cb) {
  if (f < 3 * np) {
    const int q = f / 3, c = f - 3 * q;
    return pts[3 * (int64_t)(p0 + row_idx[cb.i[q]]) + c] - pts[3 * (int64_t)(p0 + row_idx[cb.j[q]]) + c];
  }
  const int q = f - 3 * np;
  const float* ni = nrm + 3 * (int64_t)(p0 + row_idx[cb.i[q]]);
  const float* nj = nrm + 3 * (int64_t)(p0 + row_idx[cb.j[q]]);
  const float s = (ni[0] * nj[0] + ni[1] * nj[1]) + ni[2] * nj[2];
  // max(sum(n_i*n_j), sum(-n_i*n_j)): the second sum is exactly -s (negation commutes with rounding)
  return fmaxf(s, -s);
}

// Work items of a scene: first nt*head_vec "head" float4s (pair differences + normal cosines, a few dependent
// gathers each), then nt*feat_vec pure gather-copies of the feature table -- so that every wavefront (but one)
// runs a single kind of item.  blockIdx.y = scene: no per-item scene search.
// 16-byte streaming store: the 1.8 GB of encoded rows are written once and not re-read by this kernel, so they go
// out non-temporally (measured +25 % on this kernel: 2.7 -> 3.4 TB/s) instead of displacing the gathered tables.
__device__ __forceinline__ void store_stream(float* dst, const float4& v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f o;
  o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
  __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dst));
}

// KC / FV > 0: tuple size and float4s per feature row known at compile time (divisions by constants);
// 0: runtime values.  Item indices are 32-bit (a scene has < 2^31 float4 items).
template <int KC, int FV>
__global__ __launch_bounds__(256) void encode_shot_kernel(const float* __restrict__ pts,
                                                          const float* __restrict__ nrm,
                                                          const float* __restrict__ feat, int feat_dim_rt,
                                                          const int32_t* __restrict__ idx, int k_rt,
                                                          const int32_t* __restrict__ pt_off,
                                                          const int32_t* __restrict__ tup_off, ComboTable cb,
                                                          float* __restrict__ out) {
  // (tried: mapping all workgroups of a scene onto one XCD so its feature table stays in one L2 -- 10 % slower,
  //  the kernel is bound by the 1.8 GB of streaming writes, not by the gathers.  Round 2: fewer, longer-running
  //  workgroups (grid capped at 256 ... 16 per scene: 0.59 ... 0.86 ms against 0.56) and items in output order so that
  //  every wavefront stores 1 KiB of consecutive addresses with its head lanes diverging (1.38 ms) are both slower; so is
  //  one contiguous span per workgroup with four gather-copies in flight per thread (0.57 ms at best).)
  const unsigned b = blockIdx.y, bx = blockIdx.x, bps = gridDim.x;
  const int p0 = pt_off[b];
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const unsigned k = KC > 0 ? KC : (unsigned)k_rt;
  const unsigned fvec = FV > 0 ? FV : (unsigned)(feat_dim_rt >> 2);   // float4s per feature row
  const unsigned feat_dim = fvec << 2;
  const unsigned np = k * (k - 1) / 2;
  const unsigned head = 4 * np;                  // coord (3np) + normal (np) scalars
  const unsigned row_len = head + k * feat_dim;
  const unsigned head_vec = np;                  // head / 4
  const unsigned feat_vec = k * fvec;
  const unsigned n_head = (unsigned)nt * head_vec;
  const unsigned n_all = n_head + (unsigned)nt * feat_vec;
  for (unsigned v = bx * blockDim.x + threadIdx.x; v < n_all; v += bps * blockDim.x) {
    if (v >= n_head) {
      const unsigned w = v - n_head;
      const unsigned t = w / feat_vec;
      const unsigned c = w - t * feat_vec;
      const unsigned kk = c / fvec, col = (c - kk * fvec) << 2;
      const int64_t row = (int64_t)t0 + t;
      const float4 o = *reinterpret_cast<const float4*>(feat + (int64_t)(p0 + idx[row * k + kk]) * feat_dim + col);
      store_stream(out + row * row_len + head + kk * feat_dim + col, o);
    } else {
      const unsigned t = v / head_vec;
      const unsigned c = v - t * head_vec;
      const int64_t row = (int64_t)t0 + t;
      const int32_t* row_idx = idx + row * k;
      float4 o;
      o.x = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 0), cb);
      o.y = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 1), cb);
      o.z = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 2), cb);
      o.w = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 3), cb);
      store_stream(out + row * row_len + 4 * c, o);
    }
  }
}

// Half-precision feature table (BASELINE config 5: "fp16 features"): same row layout, float32 output; one item =
// 8 halfs (16-byte load) -> two streaming float4 stores.  Head items are identical to the float32 kernel's.
__global__ __launch_bounds__(256) void encode_shot_f16_kernel(const float* __restrict__ pts,
                                                              const float* __restrict__ nrm,
                                                              const __half* __restrict__ feat, int feat_dim,
                                                              const int32_t* __restrict__ idx, int k_rt,
                                                              const int32_t* __restrict__ pt_off,
                                                              const int32_t* __restrict__ tup_off, ComboTable cb,
                                                              float* __restrict__ out) {
  const unsigned b = blockIdx.y, bx = blockIdx.x, bps = gridDim.x;
  const int p0 = pt_off[b];
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const unsigned k = (unsigned)k_rt, fvec = (unsigned)feat_dim >> 3;       // 8-half groups per feature row
  const unsigned np = k * (k - 1) / 2, head = 4 * np, row_len = head + k * (unsigned)feat_dim;
  const unsigned feat_vec = k * fvec;
  const unsigned n_head = (unsigned)nt * np, n_all = n_head + (unsigned)nt * feat_vec;
  for (unsigned v = bx * blockDim.x + threadIdx.x; v < n_all; v += bps * blockDim.x) {
    if (v >= n_head) {
      const unsigned w = v - n_head, t = w / feat_vec, c = w - t * feat_vec;
      const unsigned kk = c / fvec, col = (c - kk * fvec) << 3;
      const int64_t row = (int64_t)t0 + t;
      const uint4 raw = *reinterpret_cast<const uint4*>(feat + (int64_t)(p0 + idx[row * k + kk]) * feat_dim + col);
      const __half2* h2 = reinterpret_cast<const __half2*>(&raw);
      const float2 f0 = __half22float2(h2[0]), f1 = __half22float2(h2[1]), f2 = __half22float2(h2[2]),
                   f3 = __half22float2(h2[3]);
      float* dst = out + row * row_len + head + kk * feat_dim + col;
      store_stream(dst, make_float4(f0.x, f0.y, f1.x, f1.y));
      store_stream(dst + 4, make_float4(f2.x, f2.y, f3.x, f3.y));
    } else {
      const unsigned t = v / np, c = v - t * np;
      const int64_t row = (int64_t)t0 + t;
      const int32_t* row_idx = idx + row * k;
      float4 o;
      o.x = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 0), cb);
      o.y = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 1), cb);
      o.z = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 2), cb);
      o.w = encode_scalar(pts, nrm, row_idx, p0, (int)np, (int)(4 * c + 3), cb);
      store_stream(out + row * row_len + 4 * c, o);
    }
  }
}

extern "C" int cppf_encode_tuples_shot_f16(int B, const float* pts, const float* normals, const void* feat_half,
                                           int feat_dim, const int32_t* idx, int k, const int32_t* pt_off,
                                           const int32_t* tup_off, int64_t total_tuples, float* out, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && normals && feat_half && idx && pt_off && tup_off && out);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  CPPF_CHECK_ARG(feat_dim > 0 && feat_dim % 8 == 0);
  if (total_tuples <= 0) return CPPF_OK;
  const int np = k * (k - 1) / 2;
  const int64_t per_scene = (total_tuples + B - 1) / B * (np + k * (feat_dim / 8));
  // item ids are 32-bit per scene and the batch may be ragged: bound by the worst case (one scene holds every tuple)
  if (total_tuples * (int64_t)(np + k * (feat_dim / 8)) >= 0x7fffffffLL) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_encode_tuples_shot_f16: %lld tuples x %d items exceed 2^31; split the batch",
             (long long)total_tuples, np + k * (feat_dim / 8));
    return CPPF_EUNSUPPORTED;
  }
  int64_t bx = (per_scene + 255) / 256;
  if (bx > 4096) bx = 4096;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(encode_shot_f16_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts, normals,
                     (const __half*)feat_half, feat_dim, idx, k, pt_off, tup_off, make_combos(k), out);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_encode_tuples_shot(int B, const float* pts, const float* normals, const float* feat,
                                       int feat_dim, const int32_t* idx, int k, const int32_t* pt_off,
                                       const int32_t* tup_off, int64_t total_tuples, float* out, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && normals && feat && idx && pt_off && tup_off && out);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  CPPF_CHECK_ARG(feat_dim > 0 && feat_dim % 4 == 0);
  if (total_tuples <= 0) return CPPF_OK;
  const int np = k * (k - 1) / 2;
  const int64_t per_scene = (total_tuples + B - 1) / B * (np + k * (feat_dim / 4));
  int64_t bx = (per_scene + 255) / 256;
  if (bx > 4096) bx = 4096;
  if (bx < 1) bx = 1;
  // item ids are 32-bit per scene and the batch may be ragged: bound by the worst case (one scene holds every tuple)
  if (total_tuples * (int64_t)(np + k * (feat_dim / 4)) >= 0x7fffffffLL) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_encode_tuples_shot: %lld tuples x %d items exceed 2^31; split the batch",
             (long long)total_tuples, np + k * (feat_dim / 4));
    return CPPF_EUNSUPPORTED;
  }
  if (k == 5 && feat_dim == 64)
    hipLaunchKernelGGL((encode_shot_kernel<5, 16>), dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts,
                       normals, feat, feat_dim, idx, k, pt_off, tup_off, make_combos(k), out);
  else
    hipLaunchKernelGGL((encode_shot_kernel<0, 0>), dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts,
                       normals, feat, feat_dim, idx, k, pt_off, tup_off, make_combos(k), out);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// The pair-feature block of the tuple rows alone (the first 4 C(k,2) columns of encode_shot_kernel's rows, same
// arithmetic) plus the tuples' GLOBAL point indices: what cppf_reslayer_split_gather needs to read the per-point
// descriptors itself instead of having them copied into 1.8 GB of rows.
template <int KC>
__global__ __launch_bounds__(256) void encode_shot_heads_kernel(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                                const int32_t* __restrict__ idx, int k_rt,
                                                                const int32_t* __restrict__ pt_off,
                                                                const int32_t* __restrict__ tup_off, ComboTable cb,
                                                                float* __restrict__ heads, int ld,
                                                                int32_t* __restrict__ gidx) {
  // thread / tuple: its k points and normals are fetched once (the row kernel fetches them per output float4) and all
  // C(k,2) pair features leave as one contiguous 16 C(k,2)-byte run
  const unsigned b = blockIdx.y;
  const int p0 = pt_off[b];
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const int k = KC > 0 ? KC : k_rt;
  const int np = k * (k - 1) / 2;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < (unsigned)nt; t += gridDim.x * blockDim.x) {
    const int64_t row = (int64_t)t0 + t;
    float p[8][3], n[8][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < k) {
        const int g = p0 + idx[row * k + q];
        gidx[row * k + q] = g;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          p[q][c] = pts[3 * (int64_t)g + c];
          n[q][c] = nrm[3 * (int64_t)g + c];
        }
      }
    }
    float* dst = heads + row * ld;
    // columns [0, 3 np): point differences of the pairs; [3 np, 4 np): |cos| of their normals (encode_scalar's arithmetic)
    float o[4];
    if (KC == 0) {                               // runtime k: no register arrays indexed by the pair table (scratch); re-gather
      for (int f = 0; f < 4 * np; ++f) {
        o[f & 3] = encode_scalar(pts, nrm, idx + row * k, p0, np, f, cb);
        if ((f & 3) == 3) *reinterpret_cast<float4*>(dst + (f - 3)) = make_float4(o[0], o[1], o[2], o[3]);
      }
      continue;
    }
#pragma unroll
    for (int f = 0; f < 4 * np; ++f) {
      float v;
      if (f < 3 * np) {
        const int q = f / 3, c = f - 3 * q;
        v = p[cb.i[q]][c] - p[cb.j[q]][c];
      } else {
        const int q = f - 3 * np;
        const float* ni = n[cb.i[q]];
        const float* nj = n[cb.j[q]];
        const float s_ = (ni[0] * nj[0] + ni[1] * nj[1]) + ni[2] * nj[2];
        v = fmaxf(s_, -s_);
      }
      o[f & 3] = v;
      if ((f & 3) == 3) *reinterpret_cast<float4*>(dst + (f - 3)) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// The same rows for a compile-time k, moved in whole cache lines: a workgroup takes 256 consecutive tuples, reads their k x 256
// indices as one contiguous run (and writes the global indices the same way), every thread computes its tuple's 4 C(k,2)
// features into an LDS tile (rows padded to 4 C(k,2) + 4 words: the 16-byte writes of 16 neighbouring lanes fall into different banks)
// and the tile leaves as consecutive 16-byte stores -- the thread-per-tuple kernel above issues every store with a row pitch
// between lanes and every index load with a 4 k-byte pitch.
struct __attribute__((packed, aligned(4))) Row3 {
  float x, y, z;
};

// COORD: the DINO model's head block instead (train_dino.py:92): the 3 C(k,2) coordinate differences only, zero-padded to a
// multiple of 8 columns (cppf_encode_tuples_coord_heads); nrm is not read.
template <int KC, bool COORD = false>
__global__ __launch_bounds__(256) void encode_shot_heads_tile_kernel(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                                     const int32_t* __restrict__ idx,
                                                                     const int32_t* __restrict__ pt_off,
                                                                     const int32_t* __restrict__ tup_off, ComboTable cb,
                                                                     float* __restrict__ heads, int ld,
                                                                     int32_t* __restrict__ gidx) {
  constexpr int NP = KC * (KC - 1) / 2, NF = COORD ? (3 * NP + 7) / 8 * 8 : 4 * NP, PITCH = NF + 4;
  __shared__ __attribute__((aligned(16))) float s_out[256 * PITCH];
  __shared__ int32_t s_idx[256 * KC];
  const unsigned b = blockIdx.y;
  const int p0 = pt_off[b];
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x * 256; tile < nt; tile += gridDim.x * 256) {
    const int rows = min(256, nt - tile);
    const int64_t row0 = (int64_t)t0 + tile;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      const int e = tid + 256 * i;
      if (e < rows * KC) {
        const int g = p0 + idx[row0 * KC + e];
        s_idx[e] = g;
        gidx[row0 * KC + e] = g;
      }
    }
    __syncthreads();
    if (tid < rows) {
      float p[KC][3], n[KC][3];
#pragma unroll
      for (int q = 0; q < KC; ++q) {
        const int g = s_idx[tid * KC + q];
        // one 12-byte load each (the rows are 4-byte aligned): a third of the gather instructions, which is what bounds this kernel
        const Row3 pv = *reinterpret_cast<const Row3*>(pts + 3 * (int64_t)g);
        p[q][0] = pv.x; p[q][1] = pv.y; p[q][2] = pv.z;
        if (!COORD) {
          const Row3 nv = *reinterpret_cast<const Row3*>(nrm + 3 * (int64_t)g);
          n[q][0] = nv.x; n[q][1] = nv.y; n[q][2] = nv.z;
        }
      }
      float o[4];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float v;
        if (f < 3 * NP) {
          const int q = f / 3, c = f - 3 * q;
          v = p[cb.i[q]][c] - p[cb.j[q]][c];
        } else if (COORD) {
          v = 0.0f;
        } else {
          const int q = f - 3 * NP;
          const float* ni = n[cb.i[q]];
          const float* nj = n[cb.j[q]];
          const float s_ = (ni[0] * nj[0] + ni[1] * nj[1]) + ni[2] * nj[2];
          v = fmaxf(s_, -s_);
        }
        o[f & 3] = v;
        if ((f & 3) == 3) *reinterpret_cast<float4*>(&s_out[tid * PITCH + (f - 3)]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    __syncthreads();
    for (int e = tid; e < rows * (NF / 4); e += 256) {
      const int r = e / (NF / 4), c4 = e - r * (NF / 4);
      *reinterpret_cast<float4*>(heads + (row0 + r) * ld + 4 * c4) = *reinterpret_cast<const float4*>(&s_out[r * PITCH + 4 * c4]);
    }
    __syncthreads();
  }
}

extern "C" int cppf_encode_tuples_shot_heads(int B, const float* pts, const float* normals, const int32_t* idx, int k,
                                             const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples,
                                             float* heads, int32_t ld_heads, int32_t* gidx, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && normals && idx && pt_off && tup_off && heads && gidx);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  const int np = k * (k - 1) / 2;
  CPPF_CHECK_ARG(ld_heads >= 4 * np && (ld_heads & 3) == 0 && (((uintptr_t)heads) & 15) == 0);
  if (total_tuples <= 0) return CPPF_OK;
  int64_t bx = ((total_tuples + B - 1) / B + 255) / 256;
  if (bx > 4096) bx = 4096;
  if (bx < 1) bx = 1;
  if (k == 5)
    hipLaunchKernelGGL(encode_shot_heads_tile_kernel<5>, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts, normals,
                       idx, pt_off, tup_off, make_combos(k), heads, ld_heads, gidx);
  else
    hipLaunchKernelGGL(encode_shot_heads_kernel<0>, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts, normals, idx,
                       k, pt_off, tup_off, make_combos(k), heads, ld_heads, gidx);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// The DINO model's head block (coordinate differences, zero-padded to a multiple of 8 columns) + global point indices: the
// inputs of cppf_reslayer_split_sumgather.  k = 5 runs the tile kernel above; other k a thread-per-tuple kernel.
__global__ __launch_bounds__(256) void encode_coord_heads_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int k,
                                                                 const int32_t* __restrict__ pt_off,
                                                                 const int32_t* __restrict__ tup_off, ComboTable cb,
                                                                 float* __restrict__ heads, int ld, int32_t* __restrict__ gidx) {
  const unsigned b = blockIdx.y;
  const int p0 = pt_off[b];
  const int t0 = tup_off[b], nt = tup_off[b + 1] - t0;
  const int np = k * (k - 1) / 2, nf = (3 * np + 7) / 8 * 8;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < (unsigned)nt; t += gridDim.x * blockDim.x) {
    const int64_t row = (int64_t)t0 + t;
    for (int q = 0; q < k; ++q) gidx[row * k + q] = p0 + idx[row * k + q];
    float* dst = heads + row * ld;
    for (int f = 0; f < nf; ++f) {
      float v = 0.0f;
      if (f < 3 * np) {
        const int q = f / 3, c = f - 3 * q;
        v = pts[3 * (int64_t)(p0 + idx[row * k + cb.i[q]]) + c] - pts[3 * (int64_t)(p0 + idx[row * k + cb.j[q]]) + c];
      }
      dst[f] = v;
    }
  }
}

extern "C" int cppf_encode_tuples_coord_heads(int B, const float* pts, const int32_t* idx, int k, const int32_t* pt_off,
                                              const int32_t* tup_off, int64_t total_tuples, float* heads, int32_t ld_heads,
                                              int32_t* gidx, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && idx && pt_off && tup_off && heads && gidx);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  const int nf = (3 * (k * (k - 1) / 2) + 7) / 8 * 8;
  CPPF_CHECK_ARG(ld_heads >= nf && (ld_heads & 3) == 0 && (((uintptr_t)heads) & 15) == 0);
  if (total_tuples <= 0) return CPPF_OK;
  int64_t bx = ((total_tuples + B - 1) / B + 255) / 256;
  if (bx > 4096) bx = 4096;
  if (bx < 1) bx = 1;
  if (k == 5)
    hipLaunchKernelGGL((encode_shot_heads_tile_kernel<5, true>), dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts,
                       (const float*)nullptr, idx, pt_off, tup_off, make_combos(k), heads, ld_heads, gidx);
  else
    hipLaunchKernelGGL(encode_coord_heads_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, pts, idx, k, pt_off,
                       tup_off, make_combos(k), heads, ld_heads, gidx);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

__device__ __forceinline__ void combo_of(int q, int k, int& i, int& j) {
  i = 0;
  int rem = q;
  while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
  j = i + 1 + rem;
}

__global__ __launch_bounds__(256) void encode_coord_kernel(int B, const float* __restrict__ pts,
                                                           const int32_t* __restrict__ idx, int k,
                                                           const int32_t* __restrict__ pt_off,
                                                           const int32_t* __restrict__ tup_off, int64_t total,
                                                           float* __restrict__ out, int out_stride) {
  const int np = k * (k - 1) / 2;
  const int per_row = 3 * np;
  const int64_t n = total * per_row;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = v / per_row;
    const int f = (int)(v - t * per_row);
    const int b = find_scene(tup_off, B, t);
    const int p0 = pt_off[b];
    const int32_t* row_idx = idx + t * k;
    int i, j;
    combo_of(f / 3, k, i, j);
    const int c = f % 3;
    out[t * out_stride + f] = pts[3 * (int64_t)(p0 + row_idx[i]) + c] - pts[3 * (int64_t)(p0 + row_idx[j]) + c];
  }
}

extern "C" int cppf_encode_tuples_coord(int B, const float* pts, const int32_t* idx, int k, const int32_t* pt_off,
                                        const int32_t* tup_off, int64_t total_tuples, float* out, int out_stride,
                                        void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && idx && pt_off && tup_off && out);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  CPPF_CHECK_ARG(out_stride >= 3 * (k * (k - 1) / 2));
  if (total_tuples <= 0) return CPPF_OK;
  const int64_t n = total_tuples * 3 * (k * (k - 1) / 2);
  const int64_t blocks = (n + 255) / 256;
  const int grid = (int)(blocks < 256 * 64 ? blocks : 256 * 64);
  hipLaunchKernelGGL(encode_coord_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, pts, idx, k, pt_off,
                     tup_off, total_tuples, out, out_stride);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// a3' (DINO model), descriptor part of prepare_tuple_inputs (train_dino.py:95-96):
//   desc_pair_transform(cat_i desc_transform(desc[idx_i])) = bias + sum_i W_i . d[idx_i]   (W = [W_0 | ... | W_{k-1}])
// so the Linear over the concatenation is a sum of k per-point products.  The caller forms the k products once per
// POINT (tables[n][i][:] = W_i d[n], k small GEMMs over N rows instead of one over T rows with K = k*D), and this kernel
// gathers and adds k rows per tuple: no [T, k*D] temporary, T/N times fewer FLOPs.  thread = (tuple, 4 channels).
__global__ __launch_bounds__(256) void encode_dino_kernel(int B, const float* __restrict__ tables, int k, int D,
                                                          const float* __restrict__ bias,
                                                          const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ pt_off,
                                                          const int32_t* __restrict__ tup_off, int64_t total,
                                                          float* __restrict__ out, int out_stride, int out_col) {
  const int dv = D >> 2;
  const int64_t n = total * dv;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = v / dv;
    const int c = (int)(v - t * dv) << 2;
    const int b = find_scene(tup_off, B, t);
    const int64_t p0 = pt_off[b];
    float4 a = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = 0; i < k; ++i) {
      const float4 r = *reinterpret_cast<const float4*>(tables + ((p0 + idx[t * k + i]) * k + i) * D + c);
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    float* o = out + t * out_stride + out_col + c;               // rows need not be 16-byte aligned (30 + 256 floats)
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  }
}

extern "C" int cppf_encode_tuples_dino(int B, const float* tables, int k, int D, const float* bias, const int32_t* idx,
                                       const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples, float* out,
                                       int out_stride, int out_col, void* stream) {
  CPPF_CHECK_ARG(B > 0 && tables && idx && pt_off && tup_off && out);
  CPPF_CHECK_ARG(k >= 1 && k <= 8 && D > 0 && D % 4 == 0 && out_col >= 0 && out_stride >= out_col + D);
  if (total_tuples <= 0) return CPPF_OK;
  const int64_t blocks = (total_tuples * (D / 4) + 255) / 256;
  const int grid = (int)(blocks < 256 * 64 ? blocks : 256 * 64);
  hipLaunchKernelGGL(encode_dino_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, tables, k, D, bias, idx,
                     pt_off, tup_off, total_tuples, out, out_stride, out_col);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// a4 + a5. bin decode fused with generate_target_pairs.
// Workgroup = 192 threads = 32 tuples x 6 coordinates: phase 1, one thread per (tuple, coord) reads its
// 32 logits (one 128-byte line), softmax-weights them and draws a bin by inverse CDF; phase 2, one thread
// per tuple assembles the pair, the metric scale and the float64 vote parameters.
// ---------------------------------------------------------------------------------------------
#define DEC_TUPLES 32
#define DEC_MAX_NB 4096

// NB > 0: bin count known at compile time, the CDF lives in registers.  NB == 0: generic bin count,
// three passes over the (L1-resident) logit line instead of a runtime-indexed array (which would spill).
// Each thread reads its own 128-byte row with 8 consecutive 16-byte loads -- one cache line per thread, every byte of
// it used.  (Fetching the block's rows with lane-consecutive loads and transposing them through padded LDS was built
// and measured: 0.39 ms instead of 0.23 ms at 64 x 20 000 tuples without a prior; the line-per-thread form already
// moves 4.9 TB/s.)
template <int NB>
__global__ __launch_bounds__(DEC_TUPLES * 6) void decode_bins_kernel(
    int B, const float* __restrict__ logits, const float* __restrict__ prior, int nb_rt,
    const float* __restrict__ uniforms, const float* __restrict__ pts, const int32_t* __restrict__ idx, int k, const int32_t* __restrict__ pt_off,
    const int32_t* __restrict__ tup_off, int64_t total, Axes9 axes, int32_t* __restrict__ bins,
    float* __restrict__ scaled, float* __restrict__ scale_out, float* __restrict__ tr, float* __restrict__ rot) {
  __shared__ int s_bin[DEC_TUPLES * 6];
  const int nb = NB > 0 ? NB : nb_rt;
  const int64_t tbase = (int64_t)blockIdx.x * DEC_TUPLES;
  {
    const int lt = threadIdx.x / 6, c = threadIdx.x - lt * 6;
    const int64_t t = tbase + lt;
    int bin = 0;
    if (t < total) {
      const float* lg = logits + (t * 6 + c) * nb;
      int cnt = 0;
      if constexpr (NB > 0) {
        float e[NB];
        const float4* lg4 = reinterpret_cast<const float4*>(lg);
#pragma unroll
        for (int j = 0; j < NB / 4; ++j) {
          const float4 v = lg4[j];
          e[4 * j + 0] = v.x; e[4 * j + 1] = v.y; e[4 * j + 2] = v.z; e[4 * j + 3] = v.w;
        }
        if (prior) {           // additive logit prior: one float32 add per logit, like `logits + prior` in torch
          const float4* pr4 = reinterpret_cast<const float4*>(prior + (t * 6 + c) * nb);
#pragma unroll
          for (int j = 0; j < NB / 4; ++j) {
            const float4 v = pr4[j];
            e[4 * j + 0] += v.x; e[4 * j + 1] += v.y; e[4 * j + 2] += v.z; e[4 * j + 3] += v.w;
          }
        }
        float m = e[0];
#pragma unroll
        for (int j = 1; j < NB; ++j) m = fmaxf(m, e[j]);
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc += softmax_exp(e[j] - m); e[j] = acc; }   // running f32 CDF
        const float target = uniforms[t * 6 + c] * acc;
#pragma unroll
        for (int j = 0; j < NB; ++j) cnt += (e[j] <= target) ? 1 : 0;        // first j with cdf[j] > target
      } else {
        const float* pr = prior ? prior + (t * 6 + c) * nb : nullptr;
        auto at = [&](int j) { return pr ? lg[j] + pr[j] : lg[j]; };
        float m = at(0);
        for (int j = 1; j < nb; ++j) m = fmaxf(m, at(j));
        float acc = 0.0f;
        for (int j = 0; j < nb; ++j) acc += softmax_exp(at(j) - m);
        const float target = uniforms[t * 6 + c] * acc;
        float run = 0.0f;
        for (int j = 0; j < nb; ++j) { run += softmax_exp(at(j) - m); cnt += (run <= target) ? 1 : 0; }
      }
      bin = cnt < nb - 1 ? cnt : nb - 1;
      if (bins) bins[t * 6 + c] = bin;
    }
    s_bin[threadIdx.x] = bin;
  }
  __syncthreads();
  if (threadIdx.x < DEC_TUPLES) {
    const int64_t t = tbase + threadIdx.x;
    if (t >= total) return;
    const int b = find_scene(tup_off, B, t);
    const int p0 = pt_off[b];
    const int i0 = idx[t * k + 0], i1 = idx[t * k + 1];
    float pr[6];
    const float den = (float)(nb - 1);
#pragma unroll
    for (int c = 0; c < 6; ++c) pr[c] = (float)s_bin[threadIdx.x * 6 + c] / den - 0.5f;   // eval.py:230
    // real pair length: np.linalg.norm(input_pairs[:,1]-input_pairs[:,0]) (un-fused f32), eval.py:233
    const float* pa = pts + 3 * (int64_t)(p0 + i0);
    const float* pb = pts + 3 * (int64_t)(p0 + i1);
    const float rx = pb[0] - pa[0], ry = pb[1] - pa[1], rz = pb[2] - pa[2];
    const float real_len = __builtin_sqrtf((rx * rx + ry * ry) + rz * rz);
    // predicted pair length: torch.norm (fused), clamp_min 1e-7, eval.py:234
    const float pred_len = norm3_fused(pr[3] - pr[0], pr[4] - pr[1], pr[5] - pr[2]);
    const float sc = real_len / fmaxf(pred_len, 1e-7f);
    float ps[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) ps[c] = pr[c] * sc;                                        // eval.py:235
    if (scale_out) scale_out[t] = sc;
    if (scaled) {
#pragma unroll
      for (int c = 0; c < 6; ++c) scaled[t * 6 + c] = ps[c];
    }
    float tr2[2], rot3[3];
    target_pair(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], 0.0, 0.0, 0.0, axes.a, tr2, rot3);
    if (tr) { tr[t * 2 + 0] = tr2[0]; tr[t * 2 + 1] = tr2[1]; }
    if (rot) { rot[t * 3 + 0] = rot3[0]; rot[t * 3 + 1] = rot3[1]; rot[t * 3 + 2] = rot3[2]; }
  }
}

// The second half of decode_bins_kernel alone: vote parameters of every tuple from already drawn bins (the first half runs as
// the epilogue of the MLP's output layer: cppf_reslayer_split_decode)
__global__ __launch_bounds__(256) void decode_targets_kernel(int B, const int32_t* __restrict__ bins, int nb,
                                                             const float* __restrict__ pts, const int32_t* __restrict__ idx, int k,
                                                             const int32_t* __restrict__ pt_off,
                                                             const int32_t* __restrict__ tup_off, int64_t total, Axes9 axes,
                                                             float* __restrict__ scaled, float* __restrict__ scale_out,
                                                             float* __restrict__ tr, float* __restrict__ rot) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = find_scene(tup_off, B, t);
  const int p0 = pt_off[b];
  const int i0 = idx[t * k + 0], i1 = idx[t * k + 1];
  float pr[6];
  const float den = (float)(nb - 1);
#pragma unroll
  for (int c = 0; c < 6; ++c) pr[c] = (float)bins[t * 6 + c] / den - 0.5f;                 // eval.py:230
  const float* pa = pts + 3 * (int64_t)(p0 + i0);
  const float* pb = pts + 3 * (int64_t)(p0 + i1);
  const float rx = pb[0] - pa[0], ry = pb[1] - pa[1], rz = pb[2] - pa[2];
  const float real_len = __builtin_sqrtf((rx * rx + ry * ry) + rz * rz);                   // eval.py:233 (un-fused f32)
  const float pred_len = norm3_fused(pr[3] - pr[0], pr[4] - pr[1], pr[5] - pr[2]);        // eval.py:234
  const float sc = real_len / fmaxf(pred_len, 1e-7f);
  float ps[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) ps[c] = pr[c] * sc;                                          // eval.py:235
  if (scale_out) scale_out[t] = sc;
  if (scaled) {
#pragma unroll
    for (int c = 0; c < 6; ++c) scaled[t * 6 + c] = ps[c];
  }
  float tr2[2], rot3[3];
  target_pair(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], 0.0, 0.0, 0.0, axes.a, tr2, rot3);
  if (tr) { tr[t * 2 + 0] = tr2[0]; tr[t * 2 + 1] = tr2[1]; }
  if (rot) { rot[t * 3 + 0] = rot3[0]; rot[t * 3 + 1] = rot3[1]; rot[t * 3 + 2] = rot3[2]; }
}

extern "C" int cppf_decode_from_bins(int B, const int32_t* bins, int nb, const float* pts, const int32_t* idx, int k,
                                     const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples, const double* h_axes,
                                     float* scaled, float* scale, float* tr, float* rot, void* stream) {
  CPPF_CHECK_ARG(B > 0 && bins && pts && idx && pt_off && tup_off && h_axes);
  CPPF_CHECK_ARG(nb >= 2 && nb <= DEC_MAX_NB && k >= 2 && k <= 8);
  if (total_tuples <= 0) return CPPF_OK;
  Axes9 ax;
  for (int i = 0; i < 9; ++i) ax.a[i] = h_axes[i];
  hipLaunchKernelGGL(decode_targets_kernel, dim3((unsigned)((total_tuples + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B,
                     bins, nb, pts, idx, k, pt_off, tup_off, total_tuples, ax, scaled, scale, tr, rot);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_decode_bins_prior(int B, const float* logits, const float* logit_prior, int nb, const float* uniforms,
                                      const float* pts, const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                                      int64_t total_tuples, const double* h_axes, int32_t* bins, float* scaled,
                                      float* scale, float* tr, float* rot, void* stream) {
  CPPF_CHECK_ARG(B > 0 && logits && uniforms && pts && idx && pt_off && tup_off && h_axes);
  CPPF_CHECK_ARG(nb >= 2 && nb <= DEC_MAX_NB);
  CPPF_CHECK_ARG(k >= 2 && k <= 8);
  if (total_tuples <= 0) return CPPF_OK;
  Axes9 ax;
  for (int i = 0; i < 9; ++i) ax.a[i] = h_axes[i];
  const int64_t blocks = (total_tuples + DEC_TUPLES - 1) / DEC_TUPLES;
  if (nb == 32)
    hipLaunchKernelGGL(decode_bins_kernel<32>, dim3((unsigned)blocks), dim3(DEC_TUPLES * 6), 0, (hipStream_t)stream,
                       B, logits, logit_prior, nb, uniforms, pts, idx, k, pt_off, tup_off, total_tuples, ax, bins, scaled, scale,
                       tr, rot);
  else
    hipLaunchKernelGGL(decode_bins_kernel<0>, dim3((unsigned)blocks), dim3(DEC_TUPLES * 6), 0, (hipStream_t)stream,
                       B, logits, logit_prior, nb, uniforms, pts, idx, k, pt_off, tup_off, total_tuples, ax, bins, scaled, scale,
                       tr, rot);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// the stable form (eval.py:225-240 has no prior)
extern "C" int cppf_decode_bins(int B, const float* logits, int nb, const float* uniforms, const float* pts,
                                const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                                int64_t total_tuples, const double* h_axes, int32_t* bins, float* scaled,
                                float* scale, float* tr, float* rot, void* stream) {
  const int st = cppf_decode_bins_prior(B, logits, nullptr, nb, uniforms, pts, idx, k, pt_off, tup_off, total_tuples, h_axes, bins,
                                        scaled, scale, tr, rot, stream);
  return st;
}

__global__ __launch_bounds__(256) void target_pairs_kernel(int B, const float* __restrict__ pairs,
                                                           const int32_t* __restrict__ tup_off, int64_t total,
                                                           Axes9 axes, const double* __restrict__ centers,
                                                           float* __restrict__ tr, float* __restrict__ rot) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    double cx = 0.0, cy = 0.0, cz = 0.0;
    if (centers) {
      const int b = find_scene(tup_off, B, t);
      cx = centers[3 * b + 0]; cy = centers[3 * b + 1]; cz = centers[3 * b + 2];
    }
    const float* p = pairs + t * 6;
    float tr2[2], rot3[3];
    target_pair(p[0], p[1], p[2], p[3], p[4], p[5], cx, cy, cz, axes.a, tr2, rot3);
    if (tr) { tr[t * 2 + 0] = tr2[0]; tr[t * 2 + 1] = tr2[1]; }
    if (rot) { rot[t * 3 + 0] = rot3[0]; rot[t * 3 + 1] = rot3[1]; rot[t * 3 + 2] = rot3[2]; }
  }
}

extern "C" int cppf_generate_target_pairs(int B, const float* pairs, const int32_t* tup_off, int64_t total_tuples,
                                          const double* h_axes, const double* centers, float* tr, float* rot,
                                          void* stream) {
  CPPF_CHECK_ARG(B > 0 && pairs && tup_off && h_axes);
  if (total_tuples <= 0) return CPPF_OK;
  Axes9 ax;
  for (int i = 0; i < 9; ++i) ax.a[i] = h_axes[i];
  const int64_t blocks = (total_tuples + 255) / 256;
  const int grid = (int)(blocks < 65535 ? blocks : 65535);
  hipLaunchKernelGGL(target_pairs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, pairs, tup_off,
                     total_tuples, ax, centers, tr, rot);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
