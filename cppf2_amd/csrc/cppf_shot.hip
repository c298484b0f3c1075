// cppf_shot.hip -- SHOT352 descriptors + normals on gfx950: replaces shot.compute / estimate_normal
// (src_shot/shot.cpp:12-100, a pybind11 wrapper over PCL 1.9.1 NormalEstimation + SHOTEstimation).
//
// One wavefront per query point (workgroup = 64 threads): the 64 lanes sweep the scene's points with
// coalesced loads, keep per-lane partial sums (float64) that are reduced with cross-lane shuffles, and
// compact the radius neighbours (ballot + prefix popcount) into an LDS list so that the expensive
// per-neighbour histogram interpolation (float64 acos/atan2) runs with every lane busy.  The 352-bin
// histogram lives in LDS and is filled with ds_add_f32.
//
// Numerics follow PCL 1.9.1's algorithm (features/impl/normal_3d.hpp, shot_lrf.hpp, shot.hpp) with the
// deviations listed in oracle/shot_oracle.c (float64 covariance about the query point, Jacobi eigen-solver,
// index-ordered neighbours).  Parity with PCL itself is UNPINNED (PCL is absent from the image).
#include "cppf_common.h"

#define SHOT_LEN 352
#define NR_BINS 10
#define MAX_SECTORS 32
#define SH_LCAP 1024      // neighbour-list capacity per query (falls back to a full rescan above it)

__device__ __forceinline__ int find_scene_pt(const int32_t* __restrict__ off, int B, int i) {
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Cyclic Jacobi, symmetric 3x3, float64; eigenvalues ascending, eigenvectors in columns.  Same operation
// sequence as oracle/shot_oracle.c:jacobi3 (only + - * / sqrt: bit-identical on CPU and GPU).
struct Eig3 {
  double w[3];
  double v[3][3];
};

#define JROT(P, Q, R)                                                                 \
  {                                                                                   \
    const double apq = a[P][Q];                                                       \
    if (!(fabs(apq) < 1e-300)) {                                                      \
      const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);                         \
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
      const double c = 1.0 / sqrt(t * t + 1.0);                                       \
      const double s = t * c;                                                         \
      a[P][P] = a[P][P] - t * apq;                                                    \
      a[Q][Q] = a[Q][Q] + t * apq;                                                    \
      a[P][Q] = 0.0;                                                                  \
      a[Q][P] = 0.0;                                                                  \
      const double arp = a[R][P], arq = a[R][Q];                                      \
      a[R][P] = c * arp - s * arq;                                                    \
      a[P][R] = a[R][P];                                                              \
      a[R][Q] = s * arp + c * arq;                                                    \
      a[Q][R] = a[R][Q];                                                              \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                 \
        const double vip = e.v[i][P], viq = e.v[i][Q];                                \
        e.v[i][P] = c * vip - s * viq;                                                \
        e.v[i][Q] = s * vip + c * viq;                                                \
      }                                                                               \
    }                                                                                 \
  }

__device__ __forceinline__ void eig_swap_cols(Eig3& e, double d[3], int i, int j) {
  if (d[i] > d[j]) {
    const double t = d[i]; d[i] = d[j]; d[j] = t;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const double u = e.v[r][i]; e.v[r][i] = e.v[r][j]; e.v[r][j] = u; }
  }
}

__device__ Eig3 jacobi3(const double c[6]) {
  double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  Eig3 e;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) e.v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    JROT(0, 1, 2)
    JROT(0, 2, 1)
    JROT(1, 2, 0)
  }
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  // stable bubble order (0,1),(1,2),(0,1) == the oracle's index bubble sort
  eig_swap_cols(e, d, 0, 1);
  eig_swap_cols(e, d, 1, 2);
  eig_swap_cols(e, d, 0, 1);
  e.w[0] = d[0]; e.w[1] = d[1]; e.w[2] = d[2];
  return e;
}

__device__ __forceinline__ float sqdist3(float px, float py, float pz, float qx, float qy, float qz) {
  const float dx = px - qx, dy = py - qy, dz = pz - qz;
  return (dx * dx + dy * dy) + dz * dz;
}

// ---------------------------------------------------------------------------------------------
// normals: pcl::NormalEstimation (radius search) + flipNormalTowardsViewpoint(origin)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void normals_kernel(int B, const float* __restrict__ pts,
                                                     const int32_t* __restrict__ pt_off, float radius,
                                                     float* __restrict__ out) {
  const int qi = blockIdx.x;
  const int b = find_scene_pt(pt_off, B, qi);
  const int p0 = pt_off[b], n = pt_off[b + 1] - p0;
  const float* sp = pts + 3 * (int64_t)p0;
  const float px = pts[3 * (int64_t)qi], py = pts[3 * (int64_t)qi + 1], pz = pts[3 * (int64_t)qi + 2];
  const float r2 = radius * radius;
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  for (int j = threadIdx.x; j < n; j += 64) {
    const float qx = sp[3 * j], qy = sp[3 * j + 1], qz = sp[3 * j + 2];
    if (sqdist3(px, py, pz, qx, qy, qz) < r2) {
      const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
      s[0] += x * x; s[1] += x * y; s[2] += x * z; s[3] += y * y; s[4] += y * z; s[5] += z * z;
      s[6] += x; s[7] += y; s[8] += z;
      ++cnt;
    }
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) s[c] = wave_sum(s[c]);
  cnt = wave_sum_i(cnt);
  if (threadIdx.x != 0) return;
  float* o = out + 3 * (int64_t)qi;
  if (cnt < 3) { o[0] = o[1] = o[2] = NAN; return; }
  const double inv = 1.0 / (double)cnt;
  const double mx = s[6] * inv, my = s[7] * inv, mz = s[8] * inv;
  const double cov[6] = {s[0] * inv - mx * mx, s[1] * inv - mx * my, s[2] * inv - mx * mz,
                         s[3] * inv - my * my, s[4] * inv - my * mz, s[5] * inv - mz * mz};
  const Eig3 e = jacobi3(cov);
  double nx = e.v[0][0], ny = e.v[1][0], nz = e.v[2][0];
  const double ct = (0.0 - (double)px) * nx + (0.0 - (double)py) * ny + (0.0 - (double)pz) * nz;
  if (ct < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
  o[0] = (float)nx; o[1] = (float)ny; o[2] = (float)nz;
}

// ---------------------------------------------------------------------------------------------
// SHOT352
// ---------------------------------------------------------------------------------------------
#define RAD_45 0.78539816339744830961566084581988
#define RAD_90 1.5707963267948966192313216916398
#define RAD_135 2.3561944901923449288469825374596
#define RAD_PI_7_8 2.7488935718910690836548129603691

__device__ __forceinline__ void shot_accumulate(float px, float py, float pz, float qx, float qy, float qz, float d2,
                                                float nqx, float nqy, float nqz, const float* rf, double radius,
                                                float* shot) {
  if (!isfinite(nqx) || !isfinite(nqy) || !isfinite(nqz)) return;
  double cosd = (double)((nqx * rf[6] + nqy * rf[7]) + nqz * rf[8]);
  if (cosd > 1.0) cosd = 1.0;
  if (cosd < -1.0) cosd = -1.0;
  double bin_distance = ((1.0 + cosd) * NR_BINS) / 2;
  const double distance = sqrt((double)d2);
  if (fabs(distance) < 1e-15) return;
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  double xf = (double)((dx * rf[0] + dy * rf[1]) + dz * rf[2]);
  double yf = (double)((dx * rf[3] + dy * rf[4]) + dz * rf[5]);
  double zf = (double)((dx * rf[6] + dy * rf[7]) + dz * rf[8]);
  if (fabs(yf) < 1e-30) yf = 0;
  if (fabs(xf) < 1e-30) xf = 0;
  if (fabs(zf) < 1e-30) zf = 0;
  const double r12 = radius / 2.0, r14 = radius / 4.0, r34 = radius * 3.0 / 4.0;
  const int bit4 = ((yf > 0) || ((yf == 0.0) && (xf < 0))) ? 1 : 0;
  const int bit3 = ((xf > 0) || ((xf == 0.0) && (yf > 0))) ? !bit4 : bit4;
  int desc_index = ((bit4 << 3) + (bit3 << 2)) << 1;
  if ((xf * yf > 0) || (xf == 0.0))
    desc_index += (fabs(xf) >= fabs(yf)) ? 0 : 4;
  else
    desc_index += (fabs(xf) > fabs(yf)) ? 4 : 0;
  desc_index += zf > 0 ? 1 : 0;
  desc_index += (distance > r12) ? 2 : 0;
  const int step_index = (int)floor(bin_distance + 0.5);
  const int volume_index = desc_index * (NR_BINS + 1);
  bin_distance -= step_index;
  double w = 1.0 - fabs(bin_distance);
  if (bin_distance > 0)
    atomicAdd(&shot[volume_index + ((step_index + 1) % NR_BINS)], (float)bin_distance);
  else
    atomicAdd(&shot[volume_index + ((step_index - 1 + NR_BINS) % NR_BINS)], -(float)bin_distance);
  if (distance > r12) {
    const double rd = (distance - r34) / r12;
    if (distance > r34) w += 1 - rd;
    else { w += 1 + rd; atomicAdd(&shot[(desc_index - 2) * (NR_BINS + 1) + step_index], -(float)rd); }
  } else {
    const double rd = (distance - r14) / r12;
    if (distance < r14) w += 1 + rd;
    else { w += 1 - rd; atomicAdd(&shot[(desc_index + 2) * (NR_BINS + 1) + step_index], (float)rd); }
  }
  double inc_cos = zf / distance;
  if (inc_cos < -1.0) inc_cos = -1.0;
  if (inc_cos > 1.0) inc_cos = 1.0;
  const double inc = acos(inc_cos);
  if (inc > RAD_90 || (fabs(inc - RAD_90) < 1e-30 && zf <= 0)) {
    const double id = (inc - RAD_135) / RAD_90;
    if (inc > RAD_135) w += 1 - id;
    else { w += 1 + id; atomicAdd(&shot[(desc_index + 1) * (NR_BINS + 1) + step_index], -(float)id); }
  } else {
    const double id = (inc - RAD_45) / RAD_90;
    if (inc < RAD_45) w += 1 + id;
    else { w += 1 - id; atomicAdd(&shot[(desc_index - 1) * (NR_BINS + 1) + step_index], (float)id); }
  }
  if (yf != 0.0 || xf != 0.0) {
    const double az = atan2(yf, xf);
    const int sel = desc_index >> 2;
    double ad = (az - (-RAD_PI_7_8 + RAD_45 * sel)) / RAD_45;
    ad = fmax(-0.5, fmin(ad, 0.5));
    if (ad > 0) {
      w += 1 - ad;
      atomicAdd(&shot[((desc_index + 4) % MAX_SECTORS) * (NR_BINS + 1) + step_index], (float)ad);
    } else {
      w += 1 + ad;
      atomicAdd(&shot[((desc_index - 4 + MAX_SECTORS) % MAX_SECTORS) * (NR_BINS + 1) + step_index], -(float)ad);
    }
  }
  atomicAdd(&shot[volume_index + step_index], (float)w);
}

__global__ __launch_bounds__(64) void shot_kernel(int B, const float* __restrict__ pts,
                                                  const int32_t* __restrict__ pt_off, const float* __restrict__ nrm,
                                                  float radius, float* __restrict__ out_shot,
                                                  float* __restrict__ out_rf) {
  __shared__ float s_hist[SHOT_LEN];
  __shared__ int s_list[SH_LCAP];
  __shared__ float s_rf[9];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const int b = find_scene_pt(pt_off, B, qi);
  const int p0 = pt_off[b], n = pt_off[b + 1] - p0;
  const float* sp = pts + 3 * (int64_t)p0;
  const float* sn = nrm + 3 * (int64_t)p0;
  const float px = pts[3 * (int64_t)qi], py = pts[3 * (int64_t)qi + 1], pz = pts[3 * (int64_t)qi + 2];
  const float r2 = radius * radius;
  float* o = out_shot + (int64_t)SHOT_LEN * qi;

  // sweep A: neighbour list + weighted covariance of the LRF (shot_lrf.hpp)
  double cov[6] = {0, 0, 0, 0, 0, 0}, sum = 0.0;
  int valid = 0, nn = 0, m = 0;   // valid: non-coincident neighbours; nn: all in-radius (incl. self); m: list fill
  for (int base = 0; base < n; base += 64) {
    const int j = base + lane;
    bool in = false;
    float qx = 0, qy = 0, qz = 0, d2 = 0;
    if (j < n) {
      qx = sp[3 * j]; qy = sp[3 * j + 1]; qz = sp[3 * j + 2];
      d2 = sqdist3(px, py, pz, qx, qy, qz);
      in = d2 < r2;
    }
    const unsigned long long mask = __ballot(in);
    if (in) {
      const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
      if (pos < SH_LCAP) s_list[pos] = j;
      if (!(qx == px && qy == py && qz == pz)) {
        const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
        const double w = (double)radius - (double)__builtin_sqrtf(d2);
        cov[0] += w * (x * x); cov[1] += w * (x * y); cov[2] += w * (x * z);
        cov[3] += w * (y * y); cov[4] += w * (y * z); cov[5] += w * (z * z);
        sum += w;
        ++valid;
      }
    }
    m += __popcll(mask);
  }
  nn = m;
  const bool listed = (m <= SH_LCAP);
  const int M = listed ? m : n;              // candidates of the later sweeps
#pragma unroll
  for (int c = 0; c < 6; ++c) cov[c] = wave_sum(cov[c]);
  sum = wave_sum(sum);
  valid = wave_sum_i(valid);
  __syncthreads();

  bool ok = valid >= 5;
  float rf[9];
  if (ok) {
#pragma unroll
    for (int c = 0; c < 6; ++c) cov[c] /= sum;
    const Eig3 e = jacobi3(cov);
    ok = isfinite(e.w[0]) && isfinite(e.w[1]) && isfinite(e.w[2]);
    double v1[3] = {e.v[0][2], e.v[1][2], e.v[2][2]};
    double v3[3] = {e.v[0][0], e.v[1][0], e.v[2][0]};
    // sweep B: sign disambiguation
    int plus1 = 0, plus3 = 0;
    for (int base = 0; base < M; base += 64) {
      const int c = base + lane;
      bool p1 = false, p3 = false;
      if (c < M) {
        const int j = listed ? s_list[c] : c;
        const float qx = sp[3 * j], qy = sp[3 * j + 1], qz = sp[3 * j + 2];
        const bool nb = (sqdist3(px, py, pz, qx, qy, qz) < r2) && !(qx == px && qy == py && qz == pz);
        if (nb) {
          const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
          p1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) >= 0.0;
          p3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) >= 0.0;
        }
      }
      plus1 += __popcll(__ballot(p1));
      plus3 += __popcll(__ballot(p3));
    }
    plus1 = 2 * plus1 - valid;
    plus3 = 2 * plus3 - valid;
    if (plus1 == 0 || plus3 == 0) {
      // tie: the 5 neighbours around the median of the (distance, index)-sorted valid list decide
      const int med = valid / 2;
      int c1 = 0, c3 = 0;
      for (int base = 0; base < M; base += 64) {
        const int c = base + lane;
        bool h1 = false, h3 = false;
        if (c < M) {
          const int j = listed ? s_list[c] : c;
          const float qx = sp[3 * j], qy = sp[3 * j + 1], qz = sp[3 * j + 2];
          const float d2 = sqdist3(px, py, pz, qx, qy, qz);
          if ((d2 < r2) && !(qx == px && qy == py && qz == pz)) {
            int rank = 0;
            for (int c2 = 0; c2 < M; ++c2) {
              const int j2 = listed ? s_list[c2] : c2;
              const float ux = sp[3 * j2], uy = sp[3 * j2 + 1], uz = sp[3 * j2 + 2];
              const float e2 = sqdist3(px, py, pz, ux, uy, uz);
              const bool nb2 = (e2 < r2) && !(ux == px && uy == py && uz == pz);
              rank += (nb2 && (e2 < d2 || (e2 == d2 && j2 < j))) ? 1 : 0;
            }
            if (rank >= med - 2 && rank <= med + 2) {
              const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
              h1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) > 0.0;
              h3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) > 0.0;
            }
          }
        }
        c1 += __popcll(__ballot(h1));
        c3 += __popcll(__ballot(h3));
      }
      if (plus1 == 0) plus1 = (c1 < 3) ? -1 : 1;
      if (plus3 == 0) plus3 = (c3 < 3) ? -1 : 1;
    }
    if (plus1 < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
    if (plus3 < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
    rf[0] = (float)v1[0]; rf[1] = (float)v1[1]; rf[2] = (float)v1[2];
    rf[6] = (float)v3[0]; rf[7] = (float)v3[1]; rf[8] = (float)v3[2];
    rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
    rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
    rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  }
  if (!ok) {
#pragma unroll
    for (int c = 0; c < 9; ++c) rf[c] = NAN;
  }
  if (out_rf && lane < 9) {
    // rf is wave-uniform; pick component `lane` without runtime-indexing the register array
    float v = rf[0];
#pragma unroll
    for (int c = 1; c < 9; ++c) v = (lane == c) ? rf[c] : v;
    out_rf[9 * (int64_t)qi + lane] = v;
  }
  if (!ok || nn < 5) {
    for (int c = lane; c < SHOT_LEN; c += 64) o[c] = NAN;
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) s_rf[c] = rf[c];
  }
  for (int c = lane; c < SHOT_LEN; c += 64) s_hist[c] = 0.0f;
  __syncthreads();
  // sweep C: interpolated histogram (shot.hpp: interpolateSingleChannel)
  for (int base = 0; base < M; base += 64) {
    const int c = base + lane;
    if (c < M) {
      const int j = listed ? s_list[c] : c;
      const float qx = sp[3 * j], qy = sp[3 * j + 1], qz = sp[3 * j + 2];
      const float d2 = sqdist3(px, py, pz, qx, qy, qz);
      if (d2 < r2) shot_accumulate(px, py, pz, qx, qy, qz, d2, sn[3 * j], sn[3 * j + 1], sn[3 * j + 2], s_rf, (double)radius, s_hist);
    }
  }
  __syncthreads();
  double acc = 0.0;
  for (int c = lane; c < SHOT_LEN; c += 64) acc += (double)s_hist[c] * (double)s_hist[c];
  acc = sqrt(wave_sum(acc));
  const float facc = (float)acc;
  for (int c = lane; c < SHOT_LEN; c += 64) o[c] = s_hist[c] / facc;
}

extern "C" int64_t cppf_shot352_workspace_bytes(int64_t total_points) { return total_points > 0 ? 256 : 0; }

extern "C" int cppf_estimate_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                                     float normal_r, float* out_normal, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out_normal && normal_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned)total_points), dim3(64), 0, (hipStream_t)stream, B, pts, pt_off,
                     normal_r, out_normal);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_shot352(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r,
                            float shot_r, float* out_shot, float* out_normal, float* out_rf, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out_shot && out_normal && normal_r > 0.0f && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  (void)workspace; (void)workspace_bytes;
  if (total_points <= 0) return CPPF_OK;
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned)total_points), dim3(64), 0, (hipStream_t)stream, B, pts, pt_off,
                     normal_r, out_normal);
  CPPF_LAUNCH_CHECK();
  hipLaunchKernelGGL(shot_kernel, dim3((unsigned)total_points), dim3(64), 0, (hipStream_t)stream, B, pts, pt_off,
                     out_normal, shot_r, out_shot, out_rf);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_shot352_from_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                                         const float* normals, float shot_r, float* out_shot, float* out_rf,
                                         void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && normals && out_shot && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  hipLaunchKernelGGL(shot_kernel, dim3((unsigned)total_points), dim3(64), 0, (hipStream_t)stream, B, pts, pt_off,
                     normals, shot_r, out_shot, out_rf);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
