// Shared device helpers for libcppf_hip.so (gfx950 only; compiled with -ffp-contract=off so that
// every float op below rounds exactly where it is written; fused operations are explicit fmaf()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cppf_hip.h"
#include "cppf_hip_experimental.h"

#define CPPF_WAVE 64

extern thread_local char g_cppf_err[256];

#define CPPF_CHECK_ARG(cond)                                                                  \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      snprintf(g_cppf_err, sizeof(g_cppf_err), "%s: invalid argument: %s", __func__, #cond);   \
      return CPPF_EINVAL;                                                                     \
    }                                                                                         \
  } while (0)

#define CPPF_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      snprintf(g_cppf_err, sizeof(g_cppf_err), "%s: %s failed: %s", __func__, #call,           \
               hipGetErrorString(e_));                                                        \
      return CPPF_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

#define CPPF_LAUNCH_CHECK()                                                                   \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess) {                                                                   \
      snprintf(g_cppf_err, sizeof(g_cppf_err), "%s: kernel launch failed: %s", __func__,       \
               hipGetErrorString(e_));                                                        \
      return CPPF_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

struct Axes9 {
  double a[9];
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Stream layout shared with oracle/cppf_oracle.py and
// cppf2_amd/synth.py: counter = (row, word-block, scene id, stream id), key = (seed lo, seed hi).
// ---------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// ---------------------------------------------------------------------------------------------
// Geometry shared by vote_center / vote_rotation (train_dino.py:176-192, 219-232).
// The torch-CPU reference fuses exactly three things (probed; see oracle/cppf_oracle.py):
//   torch.norm over 3 -> sqrt(fma(z,z, fma(y,y, x*x)));  torch.cross -> fma(a,b, -(c*d));
//   mm with K=3 -> fma(a2,b2, fma(a1,b1, a0*b0)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float norm3_fused(float x, float y, float z) {
  return __builtin_sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

__device__ __forceinline__ float cross_term(float a, float b, float c, float d) {
  return fmaf(a, b, -(c * d));
}

struct PairFrame {
  float ux, uy, uz;   // unit pair direction (a-b)/max(|a-b|,1e-7)
  float ax, ay, az;   // first point of the pair
  float cox, coy, coz;  // perpendicular seed (un-normalised)
  float nrm;          // |a-b|
  float nco;          // |co|
};

__device__ __forceinline__ PairFrame pair_frame(const float* __restrict__ p, int i0, int i1) {
  PairFrame f;
  f.ax = p[3 * i0 + 0]; f.ay = p[3 * i0 + 1]; f.az = p[3 * i0 + 2];
  const float bx = p[3 * i1 + 0], by = p[3 * i1 + 1], bz = p[3 * i1 + 2];
  const float dx = f.ax - bx, dy = f.ay - by, dz = f.az - bz;
  f.nrm = norm3_fused(dx, dy, dz);
  const float den = fmaxf(f.nrm, 1e-7f);
  f.ux = dx / den; f.uy = dy / den; f.uz = dz / den;
  // co = (0, -u_z, u_y); if |co| < 1e-7 use (-u_y, u_x, 0)   (train_dino.py:187-189)
  f.cox = 0.0f; f.coy = -f.uz; f.coz = f.uy;
  f.nco = norm3_fused(f.cox, f.coy, f.coz);
  if (f.nco < 1e-7f) {
    f.cox = -f.uy; f.coy = f.ux; f.coz = 0.0f;
    f.nco = norm3_fused(f.cox, f.coy, f.coz);
  }
  return f;
}

// dataset.py:118-135 on one pair (a, b) and a centre, with axes a1..a3; all in the reference's dtypes.
__device__ __forceinline__ void target_pair(float ax, float ay, float az, float bx, float by, float bz,
                                            double cx, double cy, double cz, const double* __restrict__ axes,
                                            float* tr2, float* rot3) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float n = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);      // np.linalg.norm: un-fused
  const float den = n + 1e-7f;
  const double ux = (double)(dx / den), uy = (double)(dy / den), uz = (double)(dz / den);
  const double acx = (double)ax - cx, acy = (double)ay - cy, acz = (double)az - cz;
  const double proj = (acx * ux + acy * uy) + acz * uz;
  const double ox = acx - proj * ux, oy = acy - proj * uy, oz = acz - proj * uz;
  const double dist = sqrt((ox * ox + oy * oy) + oz * oz);
  if (tr2) { tr2[0] = (float)proj; tr2[1] = (float)dist; }
  if (rot3) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double d = (ux * axes[3 * a + 0] + uy * axes[3 * a + 1]) + uz * axes[3 * a + 2];
      rot3[a] = (float)acos(d);
    }
  }
}

// atan2 where an angle only selects a bin / cell (SHOT interpolation angles, rotation-vote lookup cells): octant reduction + degree-8 polynomial in t^2 (near-minimax fit of
// atan(t)/t on [0, 1], max abs error 1.3e-7 rad evaluated in float = the rounding of the result itself), one v_rcp.
// Inputs are never both zero here (magnitudes below 1e-30 were flushed and the caller tests the pair).
__device__ __forceinline__ float atan2_poly(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float s = t * t;
  float p = 0.0028340641874819994f;
  p = fmaf(p, s, -0.016005029901862144f);
  p = fmaf(p, s, 0.042587608098983765f);
  p = fmaf(p, s, -0.07495445758104324f);
  p = fmaf(p, s, 0.10636754333972931f);
  p = fmaf(p, s, -0.14202570915222168f);
  p = fmaf(p, s, 0.19992484152317047f);
  p = fmaf(p, s, -0.3333306610584259f);
  p = fmaf(p, s, 1.0f);
  float r = p * t;
  r = (ay > ax) ? 1.5707963267948966f - r : r;
  r = (x < 0.0f) ? 3.141592653589793f - r : r;
  return copysignf(r, y);
}

// exp(x) for the softmax of the bin draw (x <= 0: logit minus the row maximum; eval.py:225-227), shared by decode_bins_kernel
// and the MLP output layer's epilogue so that both draw the same bins: 2^(x log2 e) on the hardware exponential with the product
// carried in two floats (hi + lo: the rounding of x * log2(e) alone would cost 5e-6 relative at x = -100), the low part applied
// as the first-order factor.  ~1.5 ulp, 6 instructions instead of expf's ~20.
__device__ __forceinline__ float softmax_exp(float x) {
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925963033500011e-8f;      // log2(e) = hi + lo
  const float t = x * l2e_hi;
  const float r = fmaf(x, l2e_lo, fmaf(x, l2e_hi, -t));                                 // what t misses of x * log2(e)
  // (x = -inf, a masked logit, would make r = NaN: below the float32 denormal range the result is 0 like expf's)
  return x < -104.0f ? 0.0f : __builtin_amdgcn_exp2f(t) * fmaf(r, 0.693147182464599609375f, 1.0f);
}

__device__ __forceinline__ int wave_lane() { return threadIdx.x & (CPPF_WAVE - 1); }

// Wavefront idioms written with the instructions they are (hipcc's __ballot / __popcll(mask & lanes_below) / __shfl_up
// detour through a materialised 0/1 predicate, two mask ANDs and the LDS crossbar respectively).
// wave_ballot: the condition's lane mask straight from the comparison.
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// lanes_below: how many set bits of mask belong to lanes below this one (v_mbcnt_lo + v_mbcnt_hi).
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// wave_inclusive_scan_u32: prefix sums over the 64 lanes on the DPP path (row shifts inside the rows of 16, then the last
// lane of a row broadcast to the following rows): six v_add_u32 with a DPP operand, no LDS trip.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source (or outside row_mask) read `old` = 0
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
  return v;
}
// upper_half: lane l < 32 gets lane l + 32's value (v_permlane32_swap, gfx950); lanes >= 32: unspecified.  Call it with the whole
// wavefront active (inside a branch only lanes < 32 take, the upper half's registers are not read).
__device__ __forceinline__ uint32_t upper_half(uint32_t v) {
  return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1];
}
__device__ __forceinline__ double upper_half(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = upper_half((uint32_t)u), hi = upper_half((uint32_t)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// row_down: a DPP row shift of a double (ctrl 0x101 / 0x102: lane l gets lane l + 1 / l + 2 of its row of 16; lanes without a
// source get 0.0).
template <int CTRL>
__device__ __forceinline__ double row_down(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// wave_sum: the xor-butterfly sum over the 64 lanes (every lane ends with the same bits), each exchange on the cheapest path that
// pairs a lane with a holder of its partner's value: quad permutes (xor 1, xor 2), half-row and row mirrors (after the quad steps
// all lanes of a quad hold the quad's sum, so the mirror partner carries what lane ^ 4 / lane ^ 8 would), v_permlane16_swap and
// v_permlane32_swap (both results added: own + partner in either order) -- no trip through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);                 // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);                 // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);                // row_half_mirror
  v += dpp_f64<0x140>(v);                // row_mirror
  {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane16_swap((uint32_t)u, (uint32_t)u, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
    v = __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
        __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
  }
  {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane32_swap((uint32_t)u, (uint32_t)u, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
    v = __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
        __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
  }
  return v;
}
// sqrt_rn: correctly rounded float32 square root of a positive NORMAL finite x (the rounding step of the compiler's own
// expansion -- the hardware root is within 1 ulp; pick the neighbour whose residual says so -- without its denormal
// pre-scaling and zero / infinity fix-up: 9 instructions instead of 17).  Anything else goes through sqrtf.
__device__ __forceinline__ float sqrt_rn(float x) {
  if (!(x >= 1e-30f && x <= 1e30f)) return __builtin_sqrtf(x);
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
  const float rd = __builtin_fmaf(-dn, s, x), ru = __builtin_fmaf(-up, s, x);
  float r = (rd <= 0.0f) ? dn : s;
  r = (ru > 0.0f) ? up : r;
  return r;
}

// first-maximum reduction on (value, index) pairs: larger value wins, ties -> smaller index.
__device__ __forceinline__ void argmax_combine(uint32_t& v, int64_t& i, uint32_t ov, int64_t oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ void wave_argmax(uint32_t& v, int64_t& i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t ov = __shfl_xor(v, off);
    const int64_t oi = __shfl_xor(i, off);
    argmax_combine(v, i, ov, oi);
  }
}
