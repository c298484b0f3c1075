// cppf_prep.hip -- the two steps in front of the hot path, gfx950 only:
//   back-projection of a masked depth map (utils/util.py:2586-2607 + the sign flip / float32 cast of eval.py:185-189)
//   and the one-random-point-per-voxel down-sample (utils/util.py:39-46: open3d voxel trace + np.random.choice).
#include "cppf_common.h"

#define PREP_THREADS 1024

// block-wide exclusive scan of one flag per thread (1024 threads)
__device__ __forceinline__ int prep_scan_flag(bool flag, int* s_wave, int* total) {
  const unsigned long long m = __ballot(flag);
  const int lane = wave_lane(), w = threadIdx.x >> 6;
  const int within = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < PREP_THREADS / 64; ++i) {
    const int c = s_wave[i];
    if (i < w) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + within;
}

struct Mat3d {
  double m[9];
};

// One workgroup walks the image in row-major order (the order np.where returns) and compacts the masked pixels
// with positive depth.  Arithmetic as the reference: float64 rays K^-1 [u, v, 1], xyz * z / xyz_z, then (after the
// reference's double negation of x and y cancels) a cast to float32.
// REF64 (cppf_backproject64): float64 depth in, float64 points out WITH the reference's negated x and y -- utils/util.py:2586-2607
// itself, bit for bit (every operation below is the one NumPy performs, in float64, in its order).
template <typename TD, typename TO, bool REF64>
__global__ __launch_bounds__(PREP_THREADS) void backproject_kernel(const TD* __restrict__ depth,
                                                                   const uint8_t* __restrict__ mask, int H, int W,
                                                                   Mat3d kinv, int cap, TO* __restrict__ pts,
                                                                   int32_t* __restrict__ rowcol,
                                                                   int32_t* __restrict__ count) {
  __shared__ int s_wave[PREP_THREADS / 64];
  const int n = H * W;
  int written = 0;
  for (int base = 0; base < n; base += PREP_THREADS) {
    const int i = base + threadIdx.x;
    bool keep = false;
    TD z = 0;
    if (i < n) {
      z = depth[i];
      keep = mask[i] != 0 && z > 0;
    }
    int tot;
    const int pos = written + prep_scan_flag(keep, s_wave, &tot);
    if (keep && pos < cap) {
      const int r = i / W, c = i - r * W;
      const double u = (double)c, v = (double)r;
      const double x = (kinv.m[0] * u + kinv.m[1] * v) + kinv.m[2];
      const double y = (kinv.m[3] * u + kinv.m[4] * v) + kinv.m[5];
      const double w = (kinv.m[6] * u + kinv.m[7] * v) + kinv.m[8];
      const double zd = (double)z;
      const double px = x * zd / w, py = y * zd / w;
      pts[3 * pos + 0] = (TO)(REF64 ? -px : px);
      pts[3 * pos + 1] = (TO)(REF64 ? -py : py);
      pts[3 * pos + 2] = (TO)(w * zd / w);
      if (rowcol) { rowcol[2 * pos] = r; rowcol[2 * pos + 1] = c; }
    }
    written += tot;
  }
  if (threadIdx.x == 0) *count = written;
}

extern "C" int cppf_backproject(const float* depth, const uint8_t* mask, int H, int W, const double* h_kinv, int cap,
                                float* out_pts, int32_t* out_rowcol, int32_t* out_count, void* stream) {
  CPPF_CHECK_ARG(depth && mask && h_kinv && out_pts && out_count && H > 0 && W > 0 && cap > 0);
  CPPF_CHECK_ARG((int64_t)H * W < 0x7fffffffLL);
  Mat3d k;
  for (int i = 0; i < 9; ++i) k.m[i] = h_kinv[i];
  hipLaunchKernelGGL((backproject_kernel<float, float, false>), dim3(1), dim3(PREP_THREADS), 0, (hipStream_t)stream, depth, mask, H, W,
                     k, cap, out_pts, out_rowcol, out_count);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_backproject64(const double* depth, const uint8_t* mask, int H, int W, const double* h_kinv, int cap,
                                  double* out_pts, int32_t* out_rowcol, int32_t* out_count, void* stream) {
  CPPF_CHECK_ARG(depth && mask && h_kinv && out_pts && out_count && H > 0 && W > 0 && cap > 0);
  CPPF_CHECK_ARG((int64_t)H * W < 0x7fffffffLL);
  Mat3d k;
  for (int i = 0; i < 9; ++i) k.m[i] = h_kinv[i];
  hipLaunchKernelGGL((backproject_kernel<double, double, true>), dim3(1), dim3(PREP_THREADS), 0, (hipStream_t)stream, depth, mask, H,
                     W, k, cap, out_pts, out_rowcol, out_count);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// voxel down-sample: open-addressing hash of voxel keys; every voxel keeps the point with the smallest Philox
// priority (= a uniformly random member, independent of thread order); survivors are emitted in ascending point
// index by an ordered compaction, so the result is reproducible.
// ---------------------------------------------------------------------------------------------
#define VOX_EMPTY 0xffffffffffffffffull

__global__ __launch_bounds__(256) void voxel_minmax_kernel(const float* __restrict__ pts, int n, float* __restrict__ mn) {
  // single workgroup: min corner of the cloud
  __shared__ float s_m[4][3];
  float m[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = fminf(m[c], pts[3 * i + c]);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[c] = fminf(m[c], __shfl_xor(m[c], off));
  if (wave_lane() == 0)
#pragma unroll
    for (int c = 0; c < 3; ++c) s_m[threadIdx.x >> 6][c] = m[c];
  __syncthreads();
  if (threadIdx.x < 3) mn[threadIdx.x] = fminf(fminf(s_m[0][threadIdx.x], s_m[1][threadIdx.x]),
                                               fminf(s_m[2][threadIdx.x], s_m[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float* __restrict__ pts, int n,
                                                           const float* __restrict__ mn, float res, uint32_t key0,
                                                           uint32_t key1, unsigned long long* __restrict__ keys,
                                                           unsigned long long* __restrict__ best, uint32_t cap_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // voxel index = floor((p - min) / res) in float32 (the host helper cppf2_amd.geometry.downsample does the same)
  const unsigned long long vx = (unsigned long long)floorf((pts[3 * i] - mn[0]) / res);
  const unsigned long long vy = (unsigned long long)floorf((pts[3 * i + 1] - mn[1]) / res);
  const unsigned long long vz = (unsigned long long)floorf((pts[3 * i + 2] - mn[2]) / res);
  const unsigned long long key = (vx << 42) | (vy << 21) | vz;
  const Philox4 pr = philox4x32_10((uint32_t)i, 0u, 0u, 7u, key0, key1);
  const unsigned long long payload = ((unsigned long long)pr.v[0] << 32) | (uint32_t)i;
  uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & cap_mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&keys[slot], VOX_EMPTY, key);
    if (prev == VOX_EMPTY || prev == key) {
      atomicMin(&best[slot], payload);
      return;
    }
    slot = (slot + 1) & cap_mask;
  }
}

__global__ __launch_bounds__(256) void voxel_mark_kernel(const unsigned long long* __restrict__ keys,
                                                         const unsigned long long* __restrict__ best, uint32_t cap,
                                                         uint8_t* __restrict__ chosen) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < cap && keys[s] != VOX_EMPTY) chosen[(uint32_t)best[s]] = 1;
}

__global__ __launch_bounds__(PREP_THREADS) void voxel_compact_kernel(const uint8_t* __restrict__ chosen, int n,
                                                                     int32_t* __restrict__ out_idx,
                                                                     int32_t* __restrict__ out_count) {
  __shared__ int s_wave[PREP_THREADS / 64];
  int written = 0;
  for (int base = 0; base < n; base += PREP_THREADS) {
    const int i = base + threadIdx.x;
    const bool keep = (i < n) && chosen[i] != 0;
    int tot;
    const int pos = written + prep_scan_flag(keep, s_wave, &tot);
    if (keep) out_idx[pos] = i;
    written += tot;
  }
  if (threadIdx.x == 0) *out_count = written;
}

static inline uint32_t vox_capacity(int64_t n) {
  uint32_t c = 1024;
  while ((int64_t)c < 2 * n) c <<= 1;
  return c;
}

extern "C" int64_t cppf_voxel_downsample_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (int64_t)vox_capacity(n) * 16 + ((n + 255) / 256 * 256) + 256;
}

extern "C" int cppf_voxel_downsample(const float* pts, int n, float res, uint64_t seed, int32_t* out_idx,
                                     int32_t* out_count, void* workspace, int64_t workspace_bytes, void* stream) {
  CPPF_CHECK_ARG(out_count && n >= 0 && res > 0.0f && (n == 0 || (pts && out_idx)));
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    CPPF_HIP(hipMemsetAsync(out_count, 0, 4, st));
    return CPPF_OK;
  }
  CPPF_CHECK_ARG(workspace && workspace_bytes >= cppf_voxel_downsample_workspace_bytes(n));
  const uint32_t cap = vox_capacity(n);
  unsigned long long* keys = (unsigned long long*)workspace;
  unsigned long long* best = keys + cap;
  uint8_t* chosen = (uint8_t*)(best + cap);
  float* mn = (float*)(chosen + (n + 255) / 256 * 256);
  CPPF_HIP(hipMemsetAsync(keys, 0xff, (size_t)cap * 16, st));           // keys = EMPTY, best = +inf
  CPPF_HIP(hipMemsetAsync(chosen, 0, (size_t)n, st));
  hipLaunchKernelGGL(voxel_minmax_kernel, dim3(1), dim3(256), 0, st, pts, n, mn);
  hipLaunchKernelGGL(voxel_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, n, mn, res, (uint32_t)seed,
                     (uint32_t)(seed >> 32), keys, best, cap - 1);
  hipLaunchKernelGGL(voxel_mark_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, keys, best, cap, chosen);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(1), dim3(PREP_THREADS), 0, st, chosen, n, out_idx, out_count);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// DINO-branch feature plumbing: interpolate_features (dataset.py:40-59) = F.grid_sample(bilinear,
// zeros padding, align_corners=False) of the patch-token map at the keypoints' pixel centres + F.normalize over the
// channels.  One wavefront per keypoint, lanes over channels; the map is addressed through element strides, so the
// ViT's native patch-major layout [h*w, C] (coalesced) and the reference's NCHW view both work without a copy.
// ---------------------------------------------------------------------------------------------
#define IF_WAVES 4

template <typename OUT>
__global__ __launch_bounds__(IF_WAVES * 64) void interpolate_features_kernel(
    const float* __restrict__ desc, int C, int h, int w, int64_t sc, int64_t sy, int64_t sx,
    const float* __restrict__ pts, int n, float strides, int normalize, OUT* __restrict__ out) {
  extern __shared__ float s_val[];                 // [IF_WAVES][C]
  const int lane = wave_lane(), wv = threadIdx.x >> 6;
  const int i = blockIdx.x * IF_WAVES + wv;
  if (i >= n) return;
  float* val = s_val + (size_t)wv * C;
  const float px = pts[2 * i], py = pts[2 * i + 1];
  const float gx = ((px + 0.5f) / (float)w / strides) * 2.0f - 1.0f;           // dataset.py:46-47
  const float gy = ((py + 0.5f) / (float)h / strides) * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * (float)w - 1.0f) / 2.0f;                      // grid_sampler_unnormalize, align_corners=False
  const float iy = ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
  const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
  const float wt[4] = {(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)};
  const float cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
  int64_t base[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ok[q] = cx[q] >= 0.0f && cx[q] <= (float)(w - 1) && cy[q] >= 0.0f && cy[q] <= (float)(h - 1);
    base[q] = ok[q] ? (int64_t)cy[q] * sy + (int64_t)cx[q] * sx : 0;
  }
  float ss = 0.0f;
  for (int c = lane; c < C; c += 64) {
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ok[q]) v += desc[base[q] + (int64_t)c * sc] * wt[q];
    val[c] = v;
    ss += v * v;
  }
  float inv = 1.0f;
  if (normalize) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    inv = fmaxf(__builtin_sqrtf(ss), 1e-12f);                                   // F.normalize eps
  }
  OUT* o = out + (int64_t)i * C;
  for (int c = lane; c < C; c += 64) o[c] = (OUT)(normalize ? val[c] / inv : val[c]);
}

extern "C" int cppf_interpolate_features(const float* desc, int C, int h, int w, int64_t stride_c, int64_t stride_y,
                                         int64_t stride_x, const float* pts, int n, float strides, int normalize,
                                         void* out, int out_f16, void* stream) {
  CPPF_CHECK_ARG(C > 0 && h > 0 && w > 0 && n >= 0 && strides > 0.0f);
  CPPF_CHECK_ARG((int64_t)IF_WAVES * C * 4 <= 64 * 1024);
  if (n == 0) return CPPF_OK;
  CPPF_CHECK_ARG(desc && pts && out);
  const dim3 grid((n + IF_WAVES - 1) / IF_WAVES), block(IF_WAVES * 64);
  const size_t lds = (size_t)IF_WAVES * C * 4;
  if (out_f16)
    hipLaunchKernelGGL(interpolate_features_kernel<_Float16>, grid, block, lds, (hipStream_t)stream, desc, C, h, w,
                       stride_c, stride_y, stride_x, pts, n, strides, normalize, (_Float16*)out);
  else
    hipLaunchKernelGGL(interpolate_features_kernel<float>, grid, block, lds, (hipStream_t)stream, desc, C, h, w,
                       stride_c, stride_y, stride_x, pts, n, strides, normalize, (float*)out);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
