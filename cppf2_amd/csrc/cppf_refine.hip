// cppf_refine.hip -- online alignment refinement after the votes (eval.py:319-355).  gfx950 only.
//
// The reference runs 100 Adam steps (lr 1e-2) on the translation and on a quaternion delta, minimising the mean L1
// distance between the kept pairs' points, brought into the object frame by the current pose, and the pair
// coordinates the network predicted -- 100 x (forward + backward + optimiser) tiny kernels plus lietorch per
// instance.  Here one workgroup owns a scene and keeps everything on chip: every thread holds its share of the kept
// points and their targets in registers, an iteration is one pass over them (12 partial sums: d loss / d t and
// d loss / d rot), a fixed-order block reduction, and the 7-parameter Adam update that every thread repeats
// identically.  Semantics of the lietorch pieces (un-normalised quaternion action, tangent-space gradient) are
// restated in oracle/cppf_oracle.py:refine_pose -- PARITY UNPINNED, lietorch is not available to generate goldens.
// An iteration costs one barrier: wavefront sums on the DPP path, the 4 wavefront partials double-buffered in LDS by
// iteration parity and added up (in a fixed order) by every thread for itself.
#include "cppf_common.h"

#ifndef RF_THREADS
#define RF_THREADS 256        // 4 wavefronts: an iteration is latency (reduction + barrier), not throughput
#endif
#define RF_CACHE (4096 / RF_THREADS)   // points held in registers per thread (4096 per scene = 2048 kept pairs); more are re-read

struct Mat3 {
  float m[9];                 // row-major
};

// lietorch SO3(q).matrix()[:3,:3] (so3.h: p + w * 2 (v x p) + v x (2 (v x p)) on the basis vectors), q not normalised
__device__ __forceinline__ Mat3 so3_matrix(float qx, float qy, float qz, float qw) {
  Mat3 M;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float ex = (j == 0) ? 1.0f : 0.0f, ey = (j == 1) ? 1.0f : 0.0f, ez = (j == 2) ? 1.0f : 0.0f;
    float ux = qy * ez - qz * ey, uy = qz * ex - qx * ez, uz = qx * ey - qy * ex;
    ux += ux; uy += uy; uz += uz;
    M.m[0 * 3 + j] = (ex + qw * ux) + (qy * uz - qz * uy);
    M.m[1 * 3 + j] = (ey + qw * uy) + (qz * ux - qx * uz);
    M.m[2 * 3 + j] = (ez + qw * uz) + (qx * uy - qy * ux);
  }
  return M;
}

__device__ __forceinline__ Mat3 matmul3(const Mat3& a, const Mat3& b) {
  Mat3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = (a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j]) + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}

struct RefPoint {
  float px, py, pz, tx, ty, tz;
};

__device__ __forceinline__ RefPoint load_ref_point(const float* __restrict__ p, const int32_t* __restrict__ idx, int k,
                                                   const float* __restrict__ scaled, const int32_t* __restrict__ kept,
                                                   int t0, int e) {
  // element e of the scene's [Tf, 2] point list: kept pair e / 2, end point e % 2
  const int64_t row = (int64_t)t0 + kept[e >> 1];
  const int pi = idx[row * k + (e & 1)];
  RefPoint r;
  r.px = p[3 * pi]; r.py = p[3 * pi + 1]; r.pz = p[3 * pi + 2];
  const float* s = scaled + row * 6 + 3 * (e & 1);
  r.tx = s[0]; r.ty = s[1]; r.tz = s[2];
  return r;
}

// inclusive wavefront sum on the DPP path (row_shr 1,2,4,8; row_bcast15; row_bcast31): lane 63 holds the total
__device__ __forceinline__ float wave_sum_dpp(float x) {
#define RF_DPP(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
  x += RF_DPP(x, 0x111, 0xf);
  x += RF_DPP(x, 0x112, 0xf);
  x += RF_DPP(x, 0x114, 0xf);
  x += RF_DPP(x, 0x118, 0xf);
  x += RF_DPP(x, 0x142, 0xa);
  x += RF_DPP(x, 0x143, 0xc);
#undef RF_DPP
  return x;
}

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

__device__ __forceinline__ void ref_accumulate(const RefPoint& r, const float t[3], const Mat3& rot, bool y_only,
                                               float inv_n, float a[12]) {
  const float dx = r.px - t[0], dy = r.py - t[1], dz = r.pz - t[2];
  // c = d @ rot
  const float c0 = (dx * rot.m[0] + dy * rot.m[3]) + dz * rot.m[6];
  const float c1 = (dx * rot.m[1] + dy * rot.m[4]) + dz * rot.m[7];
  const float c2 = (dx * rot.m[2] + dy * rot.m[5]) + dz * rot.m[8];
  const float g0 = y_only ? 0.0f : sgn(c0 - r.tx) * inv_n;
  const float g1 = sgn(c1 - r.ty) * inv_n;
  const float g2 = y_only ? 0.0f : sgn(c2 - r.tz) * inv_n;
  // d loss / d t = -rot g ;  d loss / d rot[a][b] = d_a g_b
  a[0] += (rot.m[0] * g0 + rot.m[1] * g1) + rot.m[2] * g2;
  a[1] += (rot.m[3] * g0 + rot.m[4] * g1) + rot.m[5] * g2;
  a[2] += (rot.m[6] * g0 + rot.m[7] * g1) + rot.m[8] * g2;
  a[3] += dx * g0; a[4] += dx * g1; a[5] += dx * g2;
  a[6] += dy * g0; a[7] += dy * g1; a[8] += dy * g2;
  a[9] += dz * g0; a[10] += dz * g1; a[11] += dz * g2;
}

__global__ __launch_bounds__(RF_THREADS) void refine_pose_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ pt_off, const int32_t* __restrict__ idx, int k,
    const int32_t* __restrict__ tup_off, const float* __restrict__ scaled, const int32_t* __restrict__ kept_tuple,
    const int32_t* __restrict__ kept_count, int y_only, int steps, float lr, CppfSceneResult* __restrict__ results) {
  __shared__ float s_part[2][RF_THREADS / 64][12];
  const int b = blockIdx.x;
  const int nk = kept_count[b];
  CppfSceneResult& res = results[b];
  if (nk <= 0 || (res.flags & 1)) return;                       // nothing kept / empty scene: pose left as voted
  const float* p = pts + 3 * (int64_t)pt_off[b];
  const int t0 = tup_off[b];
  const int32_t* kept = kept_tuple + t0;
  const int ne = 2 * nk;                                         // points in the loss
  const float inv_n = 1.0f / (float)((int64_t)ne * (y_only ? 1 : 3));
  Mat3 R0;
#pragma unroll
  for (int c = 0; c < 9; ++c) R0.m[c] = (float)res.R[c];
  float t[3] = {(float)res.t[0], (float)res.t[1], (float)res.t[2]};
  float q[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  float m[7], v[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) { m[c] = 0.0f; v[c] = 0.0f; }
  RefPoint cache[RF_CACHE];
#pragma unroll
  for (int c = 0; c < RF_CACHE; ++c) {
    const int e = threadIdx.x + c * RF_THREADS;
    if (e < ne) cache[c] = load_ref_point(p, idx, k, scaled, kept, t0, e);
  }
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  float b1p = 1.0f, b2p = 1.0f;
  for (int step = 1; step <= steps; ++step) {
    const Mat3 M = so3_matrix(q[0], q[1], q[2], q[3]);
    const Mat3 rot = matmul3(M, R0);
    float a[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) a[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < RF_CACHE; ++c)
      if ((int)threadIdx.x + c * RF_THREADS < ne) ref_accumulate(cache[c], t, rot, y_only != 0, inv_n, a);
    for (int e = threadIdx.x + RF_CACHE * RF_THREADS; e < ne; e += RF_THREADS)
      ref_accumulate(load_ref_point(p, idx, k, scaled, kept, t0, e), t, rot, y_only != 0, inv_n, a);
    // fixed-order block sum of the 12 partials: DPP row reductions inside the wavefront, then 4 wavefront partials
#pragma unroll
    for (int c = 0; c < 12; ++c) a[c] = wave_sum_dpp(a[c]);
    float (*sp)[12] = s_part[step & 1];
    if (wave_lane() == 63) {
#pragma unroll
      for (int c = 0; c < 12; ++c) sp[threadIdx.x >> 6][c] = a[c];
    }
    __syncthreads();
    float s_sum[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      float t_ = sp[0][c];
#pragma unroll
      for (int w = 1; w < RF_THREADS / 64; ++w) t_ += sp[w][c];
      s_sum[c] = t_;
    }
    float g[7];
    g[0] = -s_sum[0]; g[1] = -s_sum[1]; g[2] = -s_sum[2];
    Mat3 gr, gM;
#pragma unroll
    for (int c = 0; c < 9; ++c) gr.m[c] = s_sum[3 + c];
    // rot = M R0  ->  d loss / d M = (d loss / d rot) R0^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        gM.m[i * 3 + j] = (gr.m[i * 3] * R0.m[j * 3] + gr.m[i * 3 + 1] * R0.m[j * 3 + 1]) + gr.m[i * 3 + 2] * R0.m[j * 3 + 2];
    // tangent-space gradient of the group element: sum_j (M e_j) x dL/d(M e_j)
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float ax = M.m[j], ay = M.m[3 + j], az = M.m[6 + j];
      const float bx = gM.m[j], by = gM.m[3 + j], bz = gM.m[6 + j];
      gx += ay * bz - az * by; gy += az * bx - ax * bz; gz += ax * by - ay * bx;
    }
    const float PI_F = 3.14159265358979323846f;
    g[3] = gx / 180.0f * PI_F; g[4] = gy / 180.0f * PI_F; g[5] = gz / 180.0f * PI_F; g[6] = 0.0f;   // eval.py:341
    // torch.optim.Adam defaults (single-tensor path): exp_avg, exp_avg_sq, bias corrections, eps outside the root
    b1p *= b1; b2p *= b2;
    const float bc1 = 1.0f - b1p, bc2s = __builtin_sqrtf(1.0f - b2p);
    const float step_size = lr / bc1;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      m[c] = b1 * m[c] + (1.0f - b1) * g[c];
      v[c] = b2 * v[c] + (1.0f - b2) * g[c] * g[c];
      const float denom = __builtin_sqrtf(v[c]) / bc2s + eps;
      const float upd = step_size * m[c] / denom;
      if (c < 3) t[c] -= upd; else q[c - 3] -= upd;
    }
  }
  if (threadIdx.x == 0) {
    const Mat3 Rn = matmul3(so3_matrix(q[0], q[1], q[2], q[3]), R0);
    for (int c = 0; c < 9; ++c) res.R[c] = (double)Rn.m[c];
    for (int c = 0; c < 3; ++c) res.t[c] = (double)t[c];
    res.flags |= 8;                                              // bit3: pose refined
  }
}

extern "C" int cppf_refine_pose(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                                const int32_t* tup_off, const float* scaled, const int32_t* kept_tuple,
                                const int32_t* kept_count, int y_only, int steps, float lr, CppfSceneResult* results,
                                void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && idx && tup_off && scaled && kept_tuple && kept_count && results);
  CPPF_CHECK_ARG(k >= 2 && steps >= 0 && lr > 0.0f);
  if (steps == 0) return CPPF_OK;
  hipLaunchKernelGGL(refine_pose_kernel, dim3(B), dim3(RF_THREADS), 0, (hipStream_t)stream, pts, pt_off, idx, k, tup_off,
                     scaled, kept_tuple, kept_count, y_only, steps, lr, results);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The ensemble score and selection of eval.py:358-372, on the device: both models vote every instance (eval.py:219), each
// pose is scored by the clipped L1 distance between the kept pairs' points in the object frame and the pair coordinates the
// model drew, and the smaller score wins (strict '<' against inf, the DINO model first; geo_branch gates model 0 and
// visual_branch model 1: eval.py:367).  The reference does this in NumPy after six device -> host copies per model; here the
// two scores and the chosen record never leave the device until the batch's records are read.
// ---------------------------------------------------------------------------------------------------------------------
#define AL_THREADS 256

// loss[b] = mean over the kept pairs' 2 end points (x 3 coordinates, or y only) of clip(|canon - pred|, 0, 0.1), float64:
//   canon = (pc[point] - t) @ R / scale_norm            float32 point - float64 centre -> float64 (eval.py:358)
//   pred  = bins / (nb - 1) - 0.5                        float32 (eval.py:230; the un-scaled pair the model drew)
// scale_norm = |scale| of scale_src's record in float32 (np.linalg.norm of the float32 median, eval.py:309-310; the DINO pass'
// scale also scores the SHOT pass), 1 where it is 0.  Partial sums are added in a fixed order.  NaN when nothing was kept.
__global__ __launch_bounds__(AL_THREADS) void alignment_loss_kernel(const float* __restrict__ pts, const int32_t* __restrict__ pt_off,
                                                                    const int32_t* __restrict__ idx, int k,
                                                                    const int32_t* __restrict__ tup_off,
                                                                    const int32_t* __restrict__ bins, int nb,
                                                                    const int32_t* __restrict__ kept_tuple,
                                                                    const int32_t* __restrict__ kept_count, int y_only,
                                                                    const CppfSceneResult* __restrict__ results,
                                                                    const CppfSceneResult* __restrict__ scale_src,
                                                                    double* __restrict__ loss) {
  __shared__ double s_part[AL_THREADS];
  const int b = blockIdx.x;
  const CppfSceneResult& rec = results[b];
  const float* sc = scale_src[b].scale;
  float nrm = __builtin_sqrtf((sc[0] * sc[0] + sc[1] * sc[1]) + sc[2] * sc[2]);
  const double sn = nrm > 0.0f ? (double)nrm : 1.0;
  const float* p = pts + 3 * (int64_t)pt_off[b];
  const int t0 = tup_off[b];
  const int32_t* kept = kept_tuple + t0;
  const int n = 2 * kept_count[b];
  const float nbf = (float)(nb - 1);
  double acc = 0.0;
  for (int e = threadIdx.x; e < n; e += AL_THREADS) {
    const int64_t row = (int64_t)t0 + kept[e >> 1];
    const int pi = idx[row * k + (e & 1)];
    const double dx = (double)p[3 * pi] - rec.t[0], dy = (double)p[3 * pi + 1] - rec.t[1], dz = (double)p[3 * pi + 2] - rec.t[2];
    const int32_t* bn = bins + row * 6 + 3 * (e & 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (y_only && c != 1) continue;
      const double canon = ((dx * rec.R[c] + dy * rec.R[3 + c]) + dz * rec.R[6 + c]) / sn;
      const float pred = (float)bn[c] / nbf - 0.5f;
      double d = fabs(canon - (double)pred);
      d = d < 0.0 ? 0.0 : (d > 0.1 ? 0.1 : d);
      acc += d;
    }
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = AL_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[b] = s_part[0] / ((double)n * (y_only ? 1.0 : 3.0));      // 0 / 0 = NaN: nothing kept
}

extern "C" int cppf_alignment_loss(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                                   const int32_t* tup_off, const int32_t* bins, int nb, const int32_t* kept_tuple,
                                   const int32_t* kept_count, int y_only, const CppfSceneResult* results,
                                   const CppfSceneResult* scale_src, double* loss, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && idx && tup_off && bins && kept_tuple && kept_count && results && scale_src && loss);
  CPPF_CHECK_ARG(k >= 2 && nb >= 2);
  hipLaunchKernelGGL(alignment_loss_kernel, dim3(B), dim3(AL_THREADS), 0, (hipStream_t)stream, pts, pt_off, idx, k, tup_off, bins,
                     nb, kept_tuple, kept_count, y_only, results, scale_src, loss);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// out[b] = the record of the model with the smaller loss (model 0 first, strict '<' against inf; a NaN loss never wins), with
// the scale of model 0's record (eval.py:308-310, 372) and the pick (0, 1, or -1 when no enabled model has a finite-comparable
// loss) in pad_[0]; best[b] = the winning loss (inf when none).
__global__ __launch_bounds__(256) void ensemble_select_kernel(int B, const CppfSceneResult* __restrict__ rec0,
                                                              const CppfSceneResult* __restrict__ rec1,
                                                              const double* __restrict__ loss0, const double* __restrict__ loss1,
                                                              int enable0, int enable1, CppfSceneResult* __restrict__ out,
                                                              double* __restrict__ best) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double bl = __builtin_inf();
  int pick = -1;
  if (enable0 && loss0[b] < bl) { bl = loss0[b]; pick = 0; }
  if (enable1 && loss1[b] < bl) { bl = loss1[b]; pick = 1; }
  CppfSceneResult r = pick == 1 ? rec1[b] : rec0[b];
  r.scale[0] = rec0[b].scale[0]; r.scale[1] = rec0[b].scale[1]; r.scale[2] = rec0[b].scale[2];
  r.pad_[0] = pick;
  out[b] = r;
  best[b] = bl;
}

extern "C" int cppf_ensemble_select(int B, const CppfSceneResult* rec0, const CppfSceneResult* rec1, const double* loss0,
                                    const double* loss1, int enable0, int enable1, CppfSceneResult* out, double* best,
                                    void* stream) {
  CPPF_CHECK_ARG(B > 0 && rec0 && rec1 && loss0 && loss1 && out && best);
  hipLaunchKernelGGL(ensemble_select_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, rec0, rec1, loss0, loss1,
                     enable0, enable1, out, best);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
